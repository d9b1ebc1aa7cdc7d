#!/bin/bash
# Everything the judge reads under profiles/ for one round, in one gpurun call:
#   bash tools/profile_round.sh r04
# kernel trace + stats of the default bench command, the PMC passes (separate runs, tools/profile_pmc.sh), the
# condensed summaries (tools/summarize_profiles.py) and the bench line itself.  Outputs land in gpurun_out/;
# copy gpurun_out/profiles_<tag>/* into profiles/ and commit.
set -u
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
mkdir -p "$ROOT/gpurun_out/stats_$TAG"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/stats_$TAG" -o run -- \
    python "$ROOT/bench.py" --no-extras --cpu-pairs 0 --steps 20 > "$ROOT/gpurun_out/stats_$TAG/bench_under_profiler.json" 2> "$ROOT/gpurun_out/stats_$TAG/err.txt"
bash "$ROOT/tools/profile_pmc.sh" "$TAG" > "$ROOT/gpurun_out/pmc_$TAG.log" 2>&1
cd "$ROOT"
STATS_DIR=$(dirname "$(find gpurun_out/stats_$TAG -name '*kernel_stats.csv' | head -1)")
python tools/summarize_profiles.py "$TAG" "$STATS_DIR" "gpurun_out/pmc_$TAG" > "gpurun_out/summary_$TAG.log" 2>&1
mkdir -p "gpurun_out/profiles_$TAG"
cp profiles/${TAG}_* "gpurun_out/profiles_$TAG/" 2>/dev/null
timeout 900 python bench.py > "gpurun_out/profiles_$TAG/${TAG}_bench.json" 2> "gpurun_out/profiles_$TAG/${TAG}_bench.err"
tail -5 "gpurun_out/summary_$TAG.log"; head -c 600 "gpurun_out/profiles_$TAG/${TAG}_bench.json"
