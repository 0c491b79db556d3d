mkdir -p gpurun_out/r4o
python -m pytest tests/test_pipeline_gpu.py tests/test_trials_gpu.py tests/test_params_gpu.py -x -q -m gpu > gpurun_out/r4o/t.log 2>&1 || { tail -20 gpurun_out/r4o/t.log; exit 1; }
tail -2 gpurun_out/r4o/t.log
python tools/trials.py --example synth:Cm --trials 16 > gpurun_out/r4o/single.json 2> gpurun_out/r4o/e1 || exit 1
python tools/trials.py --example synth:Cm --trials 64 --batch 64 > gpurun_out/r4o/batch.json 2> gpurun_out/r4o/e2 || exit 1
python tools/trials.py --example ycb_024_bowl --trials 64 --batch 64 > gpurun_out/r4o/ycb.json 2> gpurun_out/r4o/e3 || exit 1
python tools/trials.py --example packed_dove --trials 64 --batch 64 > gpurun_out/r4o/packed.json 2> gpurun_out/r4o/e4 || exit 1
echo done
