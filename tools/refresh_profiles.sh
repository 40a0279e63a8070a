#!/bin/bash
# One recipe for the recorded evidence of both single-GPU bench workloads: rocprofv3 kernel stats
# and PMC passes on a GPU box (tools/collect_profiles.sh), then the summaries under profiles/
# (tools/summarize_profiles.py), stamped with the kernel sources they were measured on.
# Usage (in the build container, library built): tools/refresh_profiles.sh <tag prefix, e.g. r2>
set -e
P=$1
cd "$(dirname "$0")/.."
/usr/local/graft/bin/gpurun --timeout 900 -- "bash tools/collect_profiles.sh ${P}_mont16 mont16 mont 16 && bash tools/collect_profiles.sh ${P}_goldilocks20 goldilocks20 goldilocks 20"
python3 tools/summarize_profiles.py ${P}_mont16 mont16
python3 tools/summarize_profiles.py ${P}_goldilocks20 goldilocks20
