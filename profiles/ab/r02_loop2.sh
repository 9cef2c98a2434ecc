#!/bin/bash
cd "$(dirname "$0")/../.."
R=$(pwd)
python profiles/ab/closed_loop_ab.py 2>&1 | tail -2 | sed "s/^/default (NCH 1, PF 4): /"
for f in pf1 pf2 n2pf1 n2pf2; do GSH_LIB_PATH=$R/build/variants/lib_$f.so python profiles/ab/closed_loop_ab.py 2>&1 | tail -2; done
python profiles/ab/closed_loop_ab.py 2>&1 | tail -2 | sed "s/^/default again: /"
