cd "${GRAFT_REPO_ROOT:-/root/repo}"
for w in fwd dw dx "fwd,dw" "dw,dx"; do
  DC_TURNSTILE=1 DC_TS_WHERE="$w" timeout 200 python tools/r06/skip_probe.py base > /tmp/o.txt 2>&1; rc=$?
  echo "where=$w rc=$rc $(grep -o '"ms_per_step": [0-9.]*' /tmp/o.txt)"
done
