for ns in ${NSLIST:-16 12 10 8 6 5}; do
  GMVAE_NSPLIT_TOP=$ns python bench.py --config configs4_shard --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ns', $ns, round(j['ms_per_step']*1e3,1), 'us', j['parity']['rel_err'])"
done
