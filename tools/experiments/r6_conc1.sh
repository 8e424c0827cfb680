#!/bin/bash
# round 6, call 1: the voxelizer alone as the victim (vox_stress.py) next to a Dense scorer -- which aggressor kernels, which
# victim steps, what the in-kernel trap sees
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export MI_GNINA_NO_CALL_LOCK=1
V="timeout 120 python tools/experiments/vox_stress.py --iters 5000"
echo "== the round-5 reproducer"; DIAG_CALLS=300 timeout 300 python tools/experiments/concurrency_diag.py dense_1_3,crossdock_default2018_KD_4 2>&1 | tail -2
echo "== controls"
$V --no-aggressor --label "no aggressor"
$V --aggr-path f32 --label "aggressor on fp32-MFMA"
$V --aggressor crossdock_default2018 --label "aggressor crossdock_default2018"
echo "== baseline and victim variants"
$V --label "baseline"
$V --flags 2 --label "gather once"
$V --flags 4 --label "poisoned grid"
$V --flags 6 --label "gather once + poisoned"
GPU_MAX_HW_QUEUES=4 $V --label "GPU_MAX_HW_QUEUES=4"
GPU_MAX_HW_QUEUES=2 $V --label "GPU_MAX_HW_QUEUES=2"
echo "== aggressor dissection (timing switches: aggressor results are garbage, the victim's grid is what is compared)"
$V --opt MI_GNINA_D16_DBG=12 --opt MI_GNINA_K1S_DBG=4 --opt MI_GNINA_H2_DBG=12 --label "aggressor without any LDS-DMA"
$V --opt MI_GNINA_D16_DBG=66 --opt MI_GNINA_K1S_DBG=66 --opt MI_GNINA_H2_DBG=2 --label "aggressor DMA only (no K loops, no d16/k1s epilogues)"
$V --opt MI_GNINA_D16_DBG=12 --label "no DMA in d16"
$V --opt MI_GNINA_K1S_DBG=4 --label "no DMA in k1s"
$V --opt MI_GNINA_H2_DBG=12 --label "no DMA in conv3d_h2_kernel"
$V --opt MI_GNINA_D16_DBG=12 --opt MI_GNINA_K1S_DBG=4 --label "no DMA in d16 and k1s"
$V --opt MI_GNINA_D16_PERSIST=0 --opt MI_GNINA_K1S_PERSIST=0 --label "d16 / k1s not persistent"
$V --aggr-batch 8 --label "aggressor B = 8"
echo "== other pairs"
$V --victim dense_1_3 --aggressor dense_1_3_PT_KD_3 --label "victim dense_1_3, aggressor dense_1_3_PT_KD_3"
$V --victim default2017 --label "victim default2017"
echo "== trap build"
export MI_GNINA_LIB=$R/gnina_amd/lib/variants/libmi_gnina_trap.so
$V --flags 8 --label "trap"
$V --flags 10 --label "trap, gather once"
$V --flags 8 --no-aggressor --label "trap, no aggressor"
