run() { echo "== $*"; env "$@" python tools/bench_gan.py --content --steps 30 --graph 2>&1 | tail -1 | cut -c80-130; }
echo "== eager"; python tools/bench_gan.py --content --steps 30 2>&1 | tail -1 | cut -c80-130
run A=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=64
run DEBUG_HIP_GRAPH_BATCH_SIZE=1024
echo "== eager"; python tools/bench_gan.py --content --steps 30 2>&1 | tail -1 | cut -c80-130
