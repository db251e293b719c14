cd $GRAFT_REPO_ROOT
for t in 1 2; do for p in 0 1; do
  GPU_MAX_HW_QUEUES=8 PSK_SOFT_TIME_TILED=$t PSK_SOFT_PARALLEL_FIT=$p python bench.py --mixed --steps 20 --warmup 10 --no-cpu-baseline --no-few 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('TIME_TILED=$t PFIT=$p mixed ms', round(d['roofline']['launch_ms_avg'],3), d['check']['soft_phase_bit_identical'], d['kernel_stats']['channels_tiled'])"
done; done
