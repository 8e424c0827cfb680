#!/bin/bash
# round 6, call 6: the synthetic packed-fp32 CHAIN is the victim that reproduces next to a real Dense scorer.  Is it a wait-state
# hazard between dependent packed instructions (s_nop 3 / 7 behind every one)?  Which aggressor kernels are needed?
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
P="timeout 120 python tools/experiments/pk_victim_next_to_scorer.py"
PK_MODES=4,7,8,9 $P dense_1_3 | grep -v "^quiet"
export PK_MODES=4,7
for a in default2017 crossdock_default2018 crossdock_default2018_KD_4 dense_1_3_PT_KD_3 dense; do $P $a | grep -v "^quiet"; done
$P dense_1_3 MI_GNINA_D16_DBG=2 | grep -v "^quiet"
$P dense_1_3 MI_GNINA_K1S_DBG=2 | grep -v "^quiet"
$P dense_1_3 MI_GNINA_H2_DBG=2 | grep -v "^quiet"
$P dense_1_3 MI_GNINA_D16_DBG=2 MI_GNINA_K1S_DBG=2 MI_GNINA_H2_DBG=2 | grep -v "^quiet"
$P dense_1_3 MI_GNINA_D16_DBG=64 MI_GNINA_K1S_DBG=64 | grep -v "^quiet"
$P dense_1_3 MI_GNINA_CONV_PATH=f32 | grep -v "^quiet"
