#!/bin/bash
# The 32-agent single step with and without the LDS delta-row table, same box:   gpurun -- 'bash tools/exp/step32_ab.sh'
# (bench.py's single_step_launches leg: 256 recorded steps replayed from a hipGraph, HIP events)
for envs in ${ENVS:-16384 131072}; do
  for d in ${FORMS:-1 0 1 0}; do
    MAPF_TUNE=step_delta=$d python3 bench.py --config c5 --envs $envs --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-scalar-env --no-policy-rollout --no-per-gpu-shapes 2>/dev/null | \
      python3 -c "import json,sys; d=json.load(sys.stdin); s=d['single_step_launches']; print('envs $envs MAPF_TUNE=step_delta=$d: %.3f us per launch, frac %.4f  %s' % (s['roofline']['ms_per_launch_hip_events']*1e3, s['roofline']['frac'], s['kernel'][:70]))"
  done
done
