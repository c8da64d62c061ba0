#!/bin/bash
# per-kernel durations of the global adjacency build (100k-point radius graph) under env settings given as arguments
#   bash tools/exp/csr_ab.sh DC_CSR_XCD=0 DC_CSR_XCD=1
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do
  export "$kv"
  rm -rf /tmp/prof_ab
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -- python3 $GRAFT_REPO_ROOT/tools/exp/prep_parts.py > /tmp/prep_ab.txt 2>&1
  f=$(find /tmp/prof_ab -name "*kernel_stats.csv" | head -1)
  echo "== $kv"
  grep "GraphIndex" /tmp/prep_ab.txt
  python3 -c "import csv,sys; [print(f'{r[0][:40]:42s} {float(r[3])/1000:8.1f} us') for r in csv.reader(open(sys.argv[1])) if r[0].startswith('dc::k_') and (r[0][6:10] in ('fill','coun','emit','init','scan') or r[0].startswith('dc::k_bk'))]" $f
  unset "${kv%%=*}"
done
