#!/bin/bash
# g1: a REAL-WIDTH training step around the hot path (ResNet-34-width encoder + decoder, bf16 autocast, PyTorch-ROCm owns the
# conv GEMMs) under rocprofv3: per-kernel table, share of the step in lc_* kernels vs MIOpen/hipBLASLt/elementwise, MFMA busy
# cycles of the conv kernels, eager vs GraphedLoss.  Run on the GPU box through gpurun from the repo root:
#   bash scripts/profile_g1.sh r02
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/g1_${1:-r02}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
DENSE="python3 $ROOT/examples/train_dense_ddp.py --dtype bf16 --batch 32 --width 64 --steps 14"
SPARSE="python3 $ROOT/examples/train_sparse_ddp.py --batch 256 --sparse-cnt 64 --width 64 --steps 14"
# BASELINE configs[4] at zlmo's own shape (configs/zlmo.yaml): B=32, OS8 dilated trunk + atrous pyramid, 128x128 maps, 21 code planes, stride 3 => N=1849, fp16
ZLMO="python3 $ROOT/examples/train_dense_ddp.py --zlmo --batch 32 --width 64 --steps 14"
# wall-clock step, eager vs hipGraph-replayed Loss_fn (no profiler; 40 steps: the pool shows a bimodal step time)
${DENSE/--steps 14/--steps 40} > "$OUT/dense_eager.log" 2>&1
${DENSE/--steps 14/--steps 40} --graphs > "$OUT/dense_graphs.log" 2>&1
${SPARSE/--steps 14/--steps 40} > "$OUT/sparse_eager.log" 2>&1
${SPARSE/--steps 14/--steps 40} --graphs > "$OUT/sparse_graphs.log" 2>&1
${ZLMO/--steps 14/--steps 40} > "$OUT/zlmo_eager.log" 2>&1
${ZLMO/--steps 14/--steps 80} --graphs > "$OUT/zlmo_graphs.log" 2>&1   # nine sub-sampling phases = nine captures before the replays dominate
# kernel trace + stats
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/zlmo_trace" -o zlmo -- $ZLMO > "$OUT/zlmo_trace.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/dense_trace" -o dense -- $DENSE > "$OUT/dense_trace.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/sparse_trace" -o sparse -- $SPARSE > "$OUT/sparse_trace.log" 2>&1
# MFMA busy cycles per dispatch (own pass: --pmc with --kernel-trace only)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/dense_pmc" -o dense -- $DENSE > "$OUT/dense_pmc.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sparse_pmc" -o sparse -- $SPARSE > "$OUT/sparse_pmc.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/zlmo_pmc" -o zlmo -- $ZLMO > "$OUT/zlmo_pmc.log" 2>&1
# the sharded form of the same step: two ranks on this one GPU over gloo, rank 0 under the profiler (started directly with the rendezvous in its
# environment: no launcher between rocprofv3 and python), to see the counts / finish launches of Loss_xyz_bin and the clippers' all-reduces in place
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29641 WORLD_SIZE=2
ZL2="$ROOT/examples/train_dense_ddp.py --zlmo --batch 32 --width 64 --steps 14 --backend gloo --share-gpu --report-comm"
RANK=1 LOCAL_RANK=1 python3 $ZL2 > "$OUT/zlmo_2rank_r1.log" 2>&1 &
PEER=$!
RANK=0 LOCAL_RANK=0 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/zlmo2_trace" -o zlmo2 -- python3 $ZL2 > "$OUT/zlmo_2rank_r0.log" 2>&1
wait $PEER
unset MASTER_ADDR MASTER_PORT WORLD_SIZE
cd "$ROOT"
python3 scripts/summarize_g1.py "$OUT" > "$OUT/G1_SUMMARY.md" 2>&1
find "$OUT" -name "*.csv" -size +1500k -delete
find "$OUT" -name "*.db" -delete
du -sh "$OUT"; head -60 "$OUT/G1_SUMMARY.md"
