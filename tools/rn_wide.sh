#!/bin/bash
# occu_rn at config 4 over more workgroups than one XCD offers a chain (BIOLITH_HIP_RN_K) and fewer compute waves per workgroup
# (variant libraries libbiolith_hip_rncw{3,4}.so: make variant NAME=rncw4 EXTRA=-DBL_CWAVES_RN=4).   bash tools/rn_wide.sh > out.txt
L=biolith_amd/lib
run() { echo "== $1  RN_K=${2:-unset}"; if [ -n "$2" ]; then BIOLITH_HIP_RN_K=$2 timeout 300 python tools/time_rn.py $1 2>&1; else timeout 300 python tools/time_rn.py $1 2>&1; fi; }
run $L/libbiolith_hip.so
run $L/libbiolith_hip.so 64
[ -f $L/libbiolith_hip_rncw4.so ] && { run $L/libbiolith_hip_rncw4.so 64; run $L/libbiolith_hip_rncw4.so 48; }
[ -f $L/libbiolith_hip_rncw3.so ] && { run $L/libbiolith_hip_rncw3.so 64; run $L/libbiolith_hip_rncw3.so 56; }
