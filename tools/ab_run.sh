#!/bin/bash
# A/B of library builds on ONE box: every indelope_amd/lib/ab_*.so (built here with other compiler flags or sources) takes the place of
# the library in turn; the steady-state kernel averages (tools/r6_steady.sh) of each, the shipped build first and last.
cd "$(dirname "$0")/.." || exit 1
L=indelope_amd/lib
cp $L/libindelope_hip.so $L/base.keep
one() { echo "== $1"; python3 bench.py --no-cpu --no-e2e --no-other --no-check 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   C2 bench (10 000-region batches, two in flight): %.3f M regions/s, ms per step %s' % (d['value'] / 1e6, d['blocks']['ms_per_step']))"; bash tools/r6_steady.sh 2>&1 | grep -E "k_ksw_pair|k_asm_reads|k_asm_combine3<5, false, 32|k_tally<8>|k_prepack_fast|k_fallback" | awk '{printf "   %-44s %s us\n", substr($0,1,44), $(NF-3)}'; }
one base
for f in $L/ab_*.so; do cp $f $L/libindelope_hip.so; one $(basename $f .so); done
cp $L/base.keep $L/libindelope_hip.so
one base_again
