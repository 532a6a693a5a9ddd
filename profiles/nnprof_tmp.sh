cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_loss.py tests/test_gpu_shards.py tests/test_gpu_mpv.py -x -q -m gpu 2>&1 | grep -v "^$" | tail -30
