# builds and runs tools/c_host/pair_bench.c on the GPU box: bash tools/c_host/run_pair_bench.sh [steps] [warmup]
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=$(mktemp -d)
cd $R
python -m hippie_amd.export --kind unimodal --z-dim 10 --output-size 50 --batch 512 --lr 1e-3 --resident-units 15631 --seed 42 -o $T/wave.hpm > /dev/null
python -m hippie_amd.export --kind unimodal --z-dim 10 --output-size 100 --batch 512 --lr 1e-3 --clip 1.0 --resident-units 15631 --seed 43 -o $T/time.hpm > /dev/null
gcc -std=c99 -O2 -Wall -D_POSIX_C_SOURCE=199309L -I include tools/c_host/pair_bench.c -o $T/pair_bench -L hippie_amd -lhippie_hip -Wl,-rpath,$R/hippie_amd -Wl,-rpath,/opt/rocm/lib
GPU_MAX_HW_QUEUES=8 $T/pair_bench $T/wave.hpm $T/time.hpm ${1:-300} ${2:-30}
rm -rf $T
