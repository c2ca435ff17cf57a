# round 6: is a process' slow state the memory's temperature / clock?  The pile chain's time of eight processes after each other,
# each with the device's sensors read right behind it (rocm-smi; an ordinary user may read them)
cd $GRAFT_REPO_ROOT
sensors() {
  rocm-smi --showtemp --showclocks --showpower --showperflevel 2>/dev/null | grep -i "temperature\|mclk\|sclk\|fclk\|power\|performance" | sed 's/GPU\[0\]\s*: //' | tr '\n' ';' | cut -c1-600; echo
}
echo "idle: $(sensors)"
for i in 1 2 3 4 5 6 7 8; do
  timeout 300 python tools/pile_ab.py c3 0 3 4 2>&1 | grep "variant 0" | sed "s/^/process $i: /"
  echo "   $(sensors)"
  if [ $i = 4 ]; then sleep 20; echo "   (20 s idle)"; fi
done
