for i in 1 2 3; do timeout 60 python -u tools/gpu_wave_dbg.py 200 8 > gpurun_out/wave_dbg$i.log 2>&1; echo rc=$?; tail -2 gpurun_out/wave_dbg$i.log; done
timeout 120 python -u tools/gpu_wave_dbg.py 400 64 > gpurun_out/wave_dbg4.log 2>&1; echo rc=$?; tail -2 gpurun_out/wave_dbg4.log
