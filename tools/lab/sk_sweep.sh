# round 6: the split-K planner's knobs against the per-shape table (tools/lab/shape_table.py); usage on the box: bash tools/lab/sk_sweep.sh
for cfg in "640 6 256 24" "640 6 384 24" "640 6 512 24" "1024 6 512 24"; do
  set -- $cfg
  echo "== target $1 min_chunks $2 max_tiles $3 min_k $4"
  TPGSR_XBF_SPLITK_TARGET=$1 TPGSR_XBF_SPLITK_MIN_CHUNKS=$2 TPGSR_XBF_SPLITK_MAX_TILES=$3 TPGSR_XBF_SPLITK_MIN_K=$4 timeout 300 python tools/lab/shape_table.py c3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('family ms', round(d['total']['ms'],4), 'fwd+dgrad', round(d['by_kind']['fwd+dgrad']['ms'],4))
want=['1x26 ','1x101','1x51','1x2 ','1x4 ','2x27','2x8','4x16','8x32','1x1 ','1x201','1x203']
for r in d['table']:
    if any(w in (r.get('shape') or '') for w in want) and r['kind']!='wgrad': print('   ', r['kind'], r['shape'], r['terms'], r['us_per_launch'])
"
done
