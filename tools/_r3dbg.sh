mkdir -p gpurun_out/r3dbg
AMD_SERIALIZE_KERNEL=3 timeout -k 10 200 python -X faulthandler bench.py --config C3 --steps 2 --warmup 1 --graph off --no-roofline --no-cpu-baseline > gpurun_out/r3dbg/a.out 2> gpurun_out/r3dbg/a.err && echo "eager ok" &&
AMD_SERIALIZE_KERNEL=3 timeout -k 10 200 python -X faulthandler bench.py --config C3 --steps 2 --warmup 1 --no-roofline --no-cpu-baseline > gpurun_out/r3dbg/b.out 2> gpurun_out/r3dbg/b.err && echo "graph ok" &&
AMD_SERIALIZE_KERNEL=3 timeout -k 10 200 python -X faulthandler bench.py --config C3 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r3dbg/c.out 2> gpurun_out/r3dbg/c.err && echo "roofline ok"
