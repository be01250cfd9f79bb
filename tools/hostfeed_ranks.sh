# Validator ranks sharing ONE GPU, files -> .pth (globals: 1/2/4/8 ranks; blocks: 1/4) with host CPU %: how many ranks
# the host carries per GPU (docs/history/round3.md item 9).  usage: bash tools/hostfeed_ranks.sh [out dir]   (GPU box)
set -x
cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/hostfeed}; mkdir -p $O
N=32768
for R in 1 2 4 8; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $R --master-addr 127.0.0.1 --master-port $((29600+R)) tools/sweep_ranks.py $N globals /tmp/oake_ranks_$N 2>&1 | grep "rank(s) on one GPU" | tee -a $O/hostfeed_globals.log
done
for R in 1 4; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $R --master-addr 127.0.0.1 --master-port $((29700+R)) tools/sweep_ranks.py $N blocks /tmp/oake_ranks_$N 2>&1 | grep "rank(s) on one GPU" | tee -a $O/hostfeed_blocks.log
done
nproc; free -g | head -2
