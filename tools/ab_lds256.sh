#!/bin/bash
# A/B on ONE box: config 2's named kernel (LDS tile=256, force_variant 28) in the round-2 tree fa7fc2f (extracted and built
# under tools/ab/fa7fc2f, git-ignored) against HEAD, interleaved (old, new) x REPS.  VERDICT r03 "next" item 3.
#   gpurun --timeout 900 -- 'bash tools/ab_lds256.sh 3'
set -u
REPS=${1:-3}
mkdir -p gpurun_out/ab_lds256
ARGS="--nbodies 65536 --workload cube --variant 28 --steps 250 --warmup 150 --no-cpu-baseline --no-check"
for r in $(seq 1 "$REPS"); do
  (cd tools/ab/fa7fc2f && timeout -k 10 200 python bench.py $ARGS) > gpurun_out/ab_lds256/old_$r.json 2> gpurun_out/ab_lds256/old_$r.err || { echo "old leg $r failed"; tail -3 gpurun_out/ab_lds256/old_$r.err; exit 1; }
  timeout -k 10 200 python bench.py $ARGS --no-also > gpurun_out/ab_lds256/new_$r.json 2> gpurun_out/ab_lds256/new_$r.err || { echo "new leg $r failed"; tail -3 gpurun_out/ab_lds256/new_$r.err; exit 1; }
done
python - <<'EOF'
import json, glob
for tag in ("old", "new"):
    for f in sorted(glob.glob("gpurun_out/ab_lds256/%s_*.json" % tag)):
        line = [l for l in open(f) if l.startswith("{")][-1]
        j = json.loads(line)
        print("%s %s  %s  ms/step %.4f  frac %.4f  K1 %.4f ms  K1 frac %.4f" % (tag, f[-6:-5], j["config"]["kernel_variant"], j["ms_per_step"],
              j["frac_of_fp32_roofline"], j["roofline"]["avg_launch_ms"], j["roofline"]["frac"]))
EOF
