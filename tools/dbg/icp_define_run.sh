#!/bin/bash
# Developer tool: time the team launches (and, with SWEEP_SHAPES=1, the large shapes) with every tools/dbg/sweep_<k>.so, the
# product library first and last (box drift).  One gpurun call:  gpurun -- 'bash tools/dbg/icp_define_run.sh > gpurun_out/sweep.txt'
cd "$(dirname "$0")/../.."
run() {
  python tools/dbg/team_ab.py 2>&1 | tail -1
  [ -n "${SWEEP_SHAPES:-}" ] && python tools/dbg/lib_ab.py 2>&1 | grep step | tail -1
}
echo "== product"; run
for so in $(ls tools/dbg/sweep_*.so | sort -V); do
  echo "== $(cat ${so%.so}.txt)"
  ICPFLOW_HIP_LIB=$so run
done
echo "== product (again)"; run
