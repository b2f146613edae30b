# round 6: the split-K planner's knobs against the per-shape table (tools/lab/shape_table.py); usage on the box: bash tools/lab/sk_sweep.sh
for cfg in "0 640 6 160 0" "1 640 6 160 0" "1 640 6 160 1" "1 640 6 256 1"; do
  set -- $cfg
  echo "== splitk $1 target $2 min_chunks $3 max_tiles $4 over_halo $5"
  TPGSR_XBF_SPLITK=$1 TPGSR_XBF_SPLITK_TARGET=$2 TPGSR_XBF_SPLITK_MIN_CHUNKS=$3 TPGSR_XBF_SPLITK_MAX_TILES=$4 TPGSR_XBF_SPLITK_OVER_HALO=$5 timeout 300 python tools/lab/shape_table.py c3 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('family ms', round(d['total']['ms'],4), 'fwd+dgrad', round(d['by_kind']['fwd+dgrad']['ms'],4))
want=['1x26 2048->512','1x26 2048->256','1x101 512->128','1x51 512->37','1x2 256->256','1x4 256->256','2x27 512->512','1x26 512->512 2x2','2x8 256->128','2x8 128->256','4x16','8x32','1x1 512->256','1x2 256->512','1x201 128->64','1x101 128->512','1x201 64->128']
for r in d['table']:
    if any(w in (r.get('shape') or '') for w in want) and r['kind']!='wgrad': print('   ', r['kind'], r['shape'], r['terms'], r['us_per_launch'])
"
done
