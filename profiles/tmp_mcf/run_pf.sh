#!/bin/bash
cd $(dirname $0)/../..
echo "c3: $(python profiles/mcf_timing.py 2>&1 | grep solve | awk '{printf "%s ", $2}') $(python profiles/mcf_timing.py 2>&1 | grep solve | awk '{printf "%s ", $2}')"
python profiles/tmp_mcf/c4net.py 4 2 > /tmp/c4gen.log 2>&1
python profiles/tmp_mcf/c4run.py 2>&1 | tail -n 3
