#!/bin/bash
# The host-side loaders (own npz reader, N3Tree host loader, pose loaders, PNG writer) and the CPU oracle under
# AddressSanitizer + UndefinedBehaviorSanitizer, on valid files and on mutated ones.  CPU only.  tools/sanitize/run.sh [iters]
set -e
cd "$(dirname "$0")/../.."
IT=${1:-400}
T=$(mktemp -d /tmp/rto_san.XXXXXX)
python3 - "$T" <<'PY'
import sys, numpy as np
sys.path.insert(0, ".")
from rt_octree_amd import synth
T = sys.argv[1]
t = synth.make_tree(depth_limit=5, basis_dim=9, seed=3)
t.save_npz(T + "/tree.npz")
synth.write_transforms_json(T + "/transforms_test.json", synth.orbit_poses(7))
pb = np.zeros((5, 17)); pb[:, :15] = np.random.RandomState(0).rand(5, 15); pb[:, 4] = 378; pb[:, 9] = 504; pb[:, 14] = 400; pb[:, 15] = 1; pb[:, 16] = 9
np.save(T + "/poses_bounds.npy", pb)
PY
S=rt-octree_amd/csrc
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -Iinclude -I$S \
    tools/sanitize/host_fuzz.cpp $S/host/npz.cpp $S/host/n3tree_host.cpp $S/cli/poses.cpp $S/cli/imwrite.cpp -lz -o $T/host_fuzz
UBSAN_OPTIONS=print_stacktrace=1 ASAN_OPTIONS=detect_leaks=1:allocator_may_return_null=0:max_allocation_size_mb=4096 $T/host_fuzz $T/tree.npz $T/transforms_test.json $T/poses_bounds.npy $T $IT
# the CPU oracle (plain C; the checker of every parity test) instrumented the same way, through its own known-answer tests
gcc -O1 -g -fPIC -std=gnu11 -ffp-contract=off -fno-fast-math -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer \
    -shared -o $T/liborc_san.so oracle/rto_oracle.c -lm
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" RTO_ORC_LIB=$T/liborc_san.so \
    ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    python3 -m pytest tests/test_oracle_kat.py tests/test_expectation.py -x -q -p no:cacheprovider 2>&1 | tail -n 3
rm -rf "$T"
