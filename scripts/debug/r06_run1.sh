mkdir -p gpurun_out/r06a
python -m pytest tests/test_gpu_xyz_bin_sharded.py -x -q > gpurun_out/r06a/t_split.log 2>&1; echo "split tests rc=$?"
tail -5 gpurun_out/r06a/t_split.log
timeout 600 python examples/train_dense_ddp.py --zlmo --steps 12 --batch 32 > gpurun_out/r06a/zlmo_b32.log 2>&1; echo "zlmo rc=$?"
tail -4 gpurun_out/r06a/zlmo_b32.log
timeout 600 python examples/train_dense_ddp.py --zlmo --steps 12 --batch 32 --graphs > gpurun_out/r06a/zlmo_b32_graphs.log 2>&1; echo "zlmo graphs rc=$?"
tail -3 gpurun_out/r06a/zlmo_b32_graphs.log
timeout 600 python scripts/bench_xyz_bin_routes.py --dtype f16 > gpurun_out/r06a/routes_f16.json 2> gpurun_out/r06a/routes_f16.err; cat gpurun_out/r06a/routes_f16.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 examples/train_dense_ddp.py --zlmo --steps 8 --batch 8 --backend gloo --share-gpu --report-comm > gpurun_out/r06a/zlmo_2rank_comm.log 2>&1; echo "2rank rc=$?"
tail -8 gpurun_out/r06a/zlmo_2rank_comm.log
python -m pytest tests/test_gpu_ddp_ranks.py -x -q > gpurun_out/r06a/t_ddp.log 2>&1; echo "ddp tests rc=$?"
tail -5 gpurun_out/r06a/t_ddp.log
