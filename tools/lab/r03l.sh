#!/bin/bash
# round 3, call l: per-shape table with the halo kernels on / off (x2), the NE=9 halo variant off by default, step A/B
OUT=gpurun_out/r03l; mkdir -p $OUT
export TMPDIR=/tmp
timeout 100 python tools/lab/shape_table.py > $OUT/shapes_default.json 2> $OUT/e1.log; echo "rc=$?"
TPGSR_XBF_HALO=0 TPGSR_XBF_WGRAD_HALO=0 timeout 100 python tools/lab/shape_table.py > $OUT/shapes_nohalo.json 2> $OUT/e2.log; echo "rc=$?"
TPGSR_XBF_PANEL=0 timeout 100 python tools/lab/shape_table.py > $OUT/shapes_nopanel.json 2> $OUT/e3.log; echo "rc=$?"
python - <<'PY'
import json
def load(p):
    try: return {(r.get("kind"), r.get("shape")): r for r in json.loads(open(p).read().strip().splitlines()[-1])["table"]}
    except Exception as e: print(p, e); return {}
a, b, c = load("gpurun_out/r03l/shapes_default.json"), load("gpurun_out/r03l/shapes_nohalo.json"), load("gpurun_out/r03l/shapes_nopanel.json")
print("| kind | shape | terms | n | default us | halo kernels off us | panel off us | TF (default) | frac |")
print("|---|---|---|---|---|---|---|---|---|")
for k, r in sorted(a.items(), key=lambda kv: -kv[1]["us_per_launch"] * kv[1]["launches"]):
    print(f"| {r['kind']} | {r.get('shape','')} | {r.get('terms','')} | {r['launches']} | {r['us_per_launch']} | {b.get(k,{}).get('us_per_launch','')} | {c.get(k,{}).get('us_per_launch','')} | {r.get('tflops','')} | {r.get('frac','')} |")
PY
B="python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --alt-prec none"
ms() { python -c "import json;d=json.load(open('$1'));print(d['ms_per_step'])" 2>/dev/null; }
for rep in 1 2; do
  TPGSR_XBF_HALO_NE9=1 timeout 60 $B > $OUT/a_$rep.json 2> $OUT/a_$rep.err; echo "x2, NE9 halo variant on (as before): $(ms $OUT/a_$rep.json)"
  timeout 60 $B > $OUT/b_$rep.json 2> $OUT/b_$rep.err; echo "x2, NE9 off (new default):          $(ms $OUT/b_$rep.json)"
done
