cd /root/repo
for rep in 1 2; do
echo "== default (closed-form tperi in k_small)"; python tools/latency_model_vs_w.py 2>/dev/null | grep "E=  50"
echo "== oldtperi variant"; OCTOFITTER_HIP_LIB=octofitter.jl_amd/lib/variants/liboctofitter_hip_oldtperi.so python tools/latency_model_vs_w.py 2>/dev/null | grep "E=  50"
done
python -m pytest tests/test_model.py tests/test_sweeps_gpu.py -m gpu -x -q 2>&1 | tail -3
