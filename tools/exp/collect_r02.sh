#!/bin/bash
# side artefacts of round 2 (run on the GPU box; outputs under gpurun_out/side_r02/)
out=$PWD/gpurun_out/side_r02; mkdir -p $out
if [ -f $PWD/gym-mapf_amd/gym_mapf_amd/lib/variants/libmapf_hip_stamps.so ]; then   # (make stamps + copy it there first)
  export MAPF_HIP_LIB=$PWD/gym-mapf_amd/gym_mapf_amd/lib/variants/libmapf_hip_stamps.so
  (python tools/stamp_profile.py 65536; MAPF_TUNE=k=4 python tools/stamp_profile.py 32768) 2>&1 | grep -v amdgpu.ids > $out/r02_rollout_stamps.txt
  unset MAPF_HIP_LIB
fi
: > $out/r02_configs.txt
for spec in "c3" "c4" "c4 --envs 32768" "c5" "c5 --envs 16384" "c2"; do
  python bench.py --config $spec --steps 20 --warmup 5 --no-cpu-baseline --no-scalar-env 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; s=d['single_step_launches']
print('bench.py --config $spec: %s envs/GPU x %d agents | fused rollout %.1f G agent-steps/s, roofline %.3f of 8 TB/s, %s | single-step launches %.1f G (%.3f) | parity %s' % (d['config']['envs_per_gpu'], d['config']['n_agents'], d['value']/1e9, r['frac'], r['kernel'], s['value']/1e9, s['roofline']['frac'], d['parity']['bit_exact']))" >> $out/r02_configs.txt
done
cat $out/r02_configs.txt
python tools/host_mode_rate.py 2>&1 | grep -v amdgpu.ids | tee $out/r02_host_mode.txt
python tools/bench_transitions.py 2>&1 | grep -v amdgpu.ids | tee $out/r02_transitions.txt
python tools/soak_parity.py 1536 256 2>&1 | grep -v amdgpu.ids | tail -4 | tee $out/r02_soak_parity.txt
