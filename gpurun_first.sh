set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import torch; print(torch.cuda.is_available(), torch.cuda.get_device_name(0))"
rocminfo | grep -E "Marketing|gfx" | head -4
nproc; free -g | head -2
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s 2>&1 | tail -40
