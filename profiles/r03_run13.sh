set -x
timeout -k 10 500 bash profiles/collect.sh r03m || exit 1
timeout -k 10 300 bash profiles/collect_pmc.sh r03m
