mkdir -p gpurun_out/r5a
python -m pytest tests -m gpu -x -q > gpurun_out/r5a/pytest.log 2>&1 || { tail -30 gpurun_out/r5a/pytest.log; exit 1; }; tail -2 gpurun_out/r5a/pytest.log

python bench.py --steps 20 --warmup 5 > gpurun_out/r5a/bench.json 2> gpurun_out/r5a/bench.err || { tail -5 gpurun_out/r5a/bench.err; exit 1; }
for ex in packed_dove ycb_024_bowl linemod_obj_06 synth:Cm; do
  n=${ex#synth:}
  python tools/trials.py --example $ex --trials 64 --seed 3 --streams 8 > gpurun_out/r5a/t_${n}_streams8.json 2>> gpurun_out/r5a/trials.err || exit 1
  python tools/trials.py --example $ex --trials 64 --seed 3 > gpurun_out/r5a/t_${n}_single.json 2>> gpurun_out/r5a/trials.err || exit 1
  python tools/trials.py --example $ex --trials 64 --seed 3 --batch 64 > gpurun_out/r5a/t_${n}_batch64.json 2>> gpurun_out/r5a/trials.err || exit 1
done
python tools/frame_latency.py 8 > gpurun_out/r5a/frame.json 2> gpurun_out/r5a/frame.err || exit 1
echo done
