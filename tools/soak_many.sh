#!/bin/bash
# bash tools/soak_many.sh FIRST COUNT STEP: COUNT ranges of STEP seeds from FIRST on, one pytest run each (larger runs exhaust the workers' memory)
A=${1:-100000}; N=${2:-24}; S=${3:-20000}; O=gpurun_out; mkdir -p $O
: > $O/fuzz_soak_many_${A}.txt
for i in $(seq 0 $((N-1))); do
  a=$((A + i*S)); b=$((a + S))
  RTO_SOAK_TIMEOUT=900 bash tools/fuzz_soak.sh $a $b 8 > /dev/null 2>&1
  echo "seeds $a..$b: $(tail -n 1 $O/fuzz_soak_${a}_${b}.txt)  $(grep -c -E '^(FAILED|ERROR)' $O/fuzz_soak_${a}_${b}.txt) failed" >> $O/fuzz_soak_many_${A}.txt
  grep -E '^(FAILED|ERROR)|words differ' $O/fuzz_soak_${a}_${b}.txt | head -5 >> $O/fuzz_soak_many_${A}.txt
  rm -f $O/fuzz_soak_${a}_${b}.txt
done
cat $O/fuzz_soak_many_${A}.txt
