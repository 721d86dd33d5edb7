#!/bin/bash
# SURVEY 8f rank 2 measured (VERDICT r3 task 9): the bench scene quantised (65536-entry RGB codebooks per basis function,
# n3tree.cpp:279-340's schema), rendered (a) expanded to dense fp16 at load like the reference and (b) straight from the
# codebooks (--quant-direct): footprint, shading time, frames/s, and the shading kernel's HBM bytes per hit entry.
# bash tools/bench_quant.sh TAG   (GPU box, repo root)  -> gpurun_out/TAG_quant_*
# bash tools/bench_quant.sh TAG [luminance|median_cut]   (round 6: the quantiser, synth.SynthTree.save_quant_npz)
T=${1:-r4}; QZ=${2:-luminance}; O=gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
DENSE=$(python3 -c "import bench; print(bench.tree_cache_path(bench.parse_args([])))")
if [ ! -f $DENSE ]; then  # (bench.py generates and caches the synthetic tree on its first run)
  timeout 600 python3 bench.py --steps 1 --warmup 0 --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 0 --spot-pixels 0 --no-exact-pass --no-full-pass > /dev/null 2>&1
fi
Q=/dev/shm/rto_bench_tree_quant_r1_${QZ}.npz
[ -f $Q ] || python3 tools/make_quant_tree.py $DENSE $Q --retain 1 --quantiser $QZ > $O/${T}_quant_make.txt 2>&1
B="bench.py --streams 1 --tree $Q --no-denoise --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 0 --no-exact-pass --no-full-pass --steps 4 --warmup 1"
timeout 900 python3 $B > $O/${T}_quant_bench_expanded.json 2> $O/${T}_quant_expanded.err
timeout 900 python3 $B --quant-direct > $O/${T}_quant_bench_direct.json 2> $O/${T}_quant_direct.err
for V in expanded direct; do
  X=""; [ $V = direct ] && X="--quant-direct"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_q_${V}_trace -- python3 $B $X --steps 2 --count-frames 0 > /dev/null 2>&1
  cp $(find $O/${T}_q_${V}_trace -name '*kernel_stats.csv' | head -1) $O/${T}_quant_${V}_kernel_stats.csv
  i=0
  for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $SET --output-format csv -d $O/${T}_q_${V}_pmc_$i -- python3 $B $X --steps 2 --count-frames 0 > /dev/null 2>&1
  done
  python3 tools/pmc_summarize.py $O/${T}_quant_${V}_pmc_summary.json $O/${T}_q_${V}_pmc_1 $O/${T}_q_${V}_pmc_2 $O/${T}_q_${V}_pmc_3 > /dev/null
  rm -rf $O/${T}_q_${V}_trace $O/${T}_q_${V}_pmc_[1-3]
done
python3 - $O $T $QZ <<'PY'
import json, sys
O, T = sys.argv[1], sys.argv[2]
out = {}
for v in ("expanded", "direct"):
    d = json.loads([l for l in open("%s/%s_quant_bench_%s.json" % (O, T, v)) if l.startswith("{")][-1])
    k = json.load(open("%s/%s_quant_%s_pmc_summary.json" % (O, T, v)))["kernels"]
    sk = next((k[n] for n in k if n.startswith("shade_kernel")), {})
    rf = d["roofline"]
    hits = rf["marched_units_per_frame"]["hit_entries"] * rf["frames_per_launch"]
    fetch = sk.get("FETCH_SIZE", {}).get("mean", 0) * 1024.0
    write = sk.get("WRITE_SIZE", {}).get("mean", 0) * 1024.0
    out[v] = {"quantiser": sys.argv[3] if len(sys.argv) > 3 else "luminance", "frames_per_s": d["value"], "tree_device_mb": d["config"]["tree_device_mb"],
              "shade_ms_per_launch": rf["shade_kernel_avg_launch_ms"], "traverse_ms_per_launch": rf["avg_launch_ms"],
              "frames_per_launch": rf["frames_per_launch"], "hit_entries_per_launch": hits,
              "shade_fetch_bytes_per_launch": fetch, "shade_write_bytes_per_launch": write,
              "shade_fetch_bytes_per_hit_entry": fetch / hits if hits else None,
              "shade_l2_hit_rate": (sk["TCC_HIT_sum"]["mean"] / (sk["TCC_HIT_sum"]["mean"] + sk["TCC_MISS_sum"]["mean"])) if "TCC_HIT_sum" in sk else None}
json.dump(out, open("%s/%s_quant_summary.json" % (O, T), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
