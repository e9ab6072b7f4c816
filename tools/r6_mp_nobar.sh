V=$PWD/octofitter.jl_amd/lib/variants
for r in 1 2; do for P in 4 5 6 8; do for lib in default nobar2 nobar1; do
 if [ $lib = default ]; then timeout 120 python tools/multi_planet_steps.py $P 100 2>&1 | grep "us per step"; else OCTOFITTER_HIP_LIB=$V/liboctofitter_hip_$lib.so timeout 120 python tools/multi_planet_steps.py $P 100 2>&1 | grep "us per step"; fi
done; done; done
