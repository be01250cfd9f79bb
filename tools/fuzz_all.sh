# Every randomised check on the current binary with one fresh seed (GPU box): bash tools/fuzz_all.sh [seed] > log
S=${1:-31}
cd $GRAFT_REPO_ROOT
echo "# session $(hostname)-$(date -u +%Y%m%dT%H%M%SZ): every randomised check on the final binary, seed $S"
python tools/kernel_fuzz.py 300 $S 2>&1 | tail -1
python tools/jpeg_fuzz.py 3000 $S 2>&1 | tail -1
python tools/resample_fuzz.py 150 $S 2>&1 | tail -1
python tools/blocks_fuzz.py 60 $S 2>&1 | tail -1
python tests/fuzz_encoder.py 60 $S 2>&1 | tail -1
python tests/fuzz_text.py 60 $S 2>&1 | tail -1
python tests/fuzz_pipeline.py 20 $S 2>&1 | tail -1
