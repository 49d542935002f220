for r in 1 2; do for b in 32 64 128 256; do for d in 1 0; do
DFM_EDGE_DYNAMIC=$d python bench.py --batch $b --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-line --no-c4-line --no-c5-line 2>/dev/null | python -c "
import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); m=j['roofline']['launch_mix']; print('B', j['config']['trajectories_per_gpu'], 'dynamic' if $d else 'static ', round(j['value'],1), 'full %.4f lig %.4f' % (m['avg_full_ms'], m['avg_ligand_only_ms']))"
done; done; done
