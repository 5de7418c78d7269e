set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/so -o so -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-trainer --no-dp-probe --no-profile > /tmp/so.json 2>/tmp/so.err
f=$(find /tmp/so -name "*kernel_trace.csv")
python3 $R/tools/stream_overlap.py $f 50 5 $O/stream_gantt.txt > $O/stream_overlap.txt
head -4 $O/stream_overlap.txt; cat /tmp/so.json | cut -c1-200
