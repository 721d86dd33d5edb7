#!/bin/bash
# counters of the bench workload's kernels for several builds of librto.so in one box:
# bash tools/ab_pmc.sh TAG "<bench args>" LIB [LIB ...]  -> gpurun_out/TAG_<libname>_pmc.json (+ a table on stdout)
T=$1; A=$2; shift 2
O=gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
B="bench.py $A --streams 1 --cpu-frames 0 --psnr-frames 0 --ref-loop-frames 0 --spot-pixels 0 --count-frames 0 --no-exact-pass --no-full-pass --steps 2 --warmup 1"
for L in "$@"; do
  N=$(basename $L .so)
  i=0
  for SET in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
             "SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_BRANCH SQ_INST_LEVEL_VMEM" \
             "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    i=$((i+1))
    RTO_LIB=$PWD/$L timeout 300 rocprofv3 --pmc $SET --output-format csv -d $O/${T}_${N}_p$i -- python3 $B > /dev/null 2> $O/${T}_${N}_p$i.err || tail -2 $O/${T}_${N}_p$i.err
  done
  python3 tools/pmc_summarize.py $O/${T}_${N}_pmc.json $O/${T}_${N}_p1 $O/${T}_${N}_p2 $O/${T}_${N}_p3 $O/${T}_${N}_p4 $O/${T}_${N}_p5 > /dev/null
  rm -rf $O/${T}_${N}_p[1-5] $O/${T}_${N}_p[1-5].err
done
python3 - $O $T "$@" <<'PY'
import json, os, sys
O, T, libs = sys.argv[1], sys.argv[2], sys.argv[3:]
for fam in ("render_persist", "shade_kernel", "sample_kernel"):
    for L in libs:
        N = os.path.basename(L)[:-3]
        k = json.load(open("%s/%s_%s_pmc.json" % (O, T, N)))["kernels"].get(fam)
        if not k:
            continue
        m = lambda c: k[c]["mean"] if c in k else float("nan")
        clk = m("SQ_BUSY_CYCLES") / 32.0
        print("%-15s %-22s clk %.2fM  VALU %.0fM (%.3f/clk/SIMD)  SALU %.0fM  lanes/VALU %.1f  wait_any %.2f  wait_inst %.2f  active %.2f  LDS %.1fM  VMEM rd %.1fM wr %.1fM  "
              "fetch %.2f GB  write %.2f GB  L2 hit %.2f  waves %.0f" % (
                  fam, N, clk / 1e6, m("SQ_INSTS_VALU") / 1e6, m("SQ_INSTS_VALU") / 1024 / clk, m("SQ_INSTS_SALU") / 1e6,
                  m("SQ_THREAD_CYCLES_VALU") / m("SQ_INSTS_VALU"), m("SQ_WAIT_ANY") / m("SQ_WAVE_CYCLES"), m("SQ_WAIT_INST_ANY") / m("SQ_WAVE_CYCLES"),
                  m("SQ_ACTIVE_INST_ANY") / m("SQ_WAVE_CYCLES"), m("SQ_INSTS_LDS") / 1e6, m("SQ_INSTS_VMEM_RD") / 1e6, m("SQ_INSTS_VMEM_WR") / 1e6,
                  m("FETCH_SIZE") * 1024 / 1e9, m("WRITE_SIZE") * 1024 / 1e9, m("TCC_HIT_sum") / (m("TCC_HIT_sum") + m("TCC_MISS_sum")), m("SQ_WAVES")))
PY
