#!/bin/bash
cd $(dirname $0)/../..
python profiles/tmp_mcf/c4net.py 4 2 > /tmp/c4gen.log 2>&1
AXT_MCF_DEBUG=1 python profiles/tmp_mcf/c4run.py 2>&1 | grep "leaf\|separ\|lsap\|c4 solve" | tail -n 36
for th in 32 64; do echo "threads $th: $(AXT_MCF_THREADS=$th python profiles/tmp_mcf/c4run.py 2>&1 | tail -n 2 | tr '\n' ' ')"; done
