# the length classes' own kernels against the any-length (sorted-event) kernel, same box
cd $GRAFT_REPO_ROOT
for i in 1 2; do
echo "class kernels:"; timeout 600 python tools/long_read_probe.py 2 3 5 2>&1 | grep "^x"
echo "any-length kernel:"; RALA_PILE_ANYLEN_FROM=1 timeout 600 python tools/long_read_probe.py 2 3 5 2>&1 | grep "^x"
done
