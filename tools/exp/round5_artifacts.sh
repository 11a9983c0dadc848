#!/bin/bash
# Round 5's side artefacts in one GPU call (written to gpurun_out/, copied to profiles/ afterwards):
#   gpurun --timeout 1200 -- 'bash tools/exp/round5_artifacts.sh'
O=gpurun_out
V=$PWD/gym-mapf_amd/gym_mapf_amd/lib/variants
echo "[stamps]"
(MAPF_HIP_LIB=$V/libmapf_hip_stamps.so python3 tools/stamp_profile.py 65536 c3; MAPF_HIP_LIB=$V/libmapf_hip_stamps.so python3 tools/stamp_profile.py 32768 c4) 2>&1 | grep -v amdgpu.ids > $O/r05_stamps_c3_c4s.txt
(MAPF_HIP_LIB=$V/libmapf_hip_stamps.so python3 tools/stamp_profile.py 16384 c5) 2>&1 | grep -v amdgpu.ids > $O/r05_stamps_c5.txt
echo "[step stamps, configs[4]'s share: plain / delta rows / delta rows + bitmaps]"
(MAPF_HIP_LIB=$V/libmapf_hip_step_stamps.so MAPF_TUNE=step_delta=0 python3 tools/step_stamps.py 16384 graph c5; MAPF_HIP_LIB=$V/libmapf_hip_step_stamps.so MAPF_TUNE=step_delta=2,bitmap_pairs=0 python3 tools/step_stamps.py 16384 graph c5; MAPF_HIP_LIB=$V/libmapf_hip_step_stamps.so MAPF_TUNE=step_delta=2 python3 tools/step_stamps.py 16384 graph c5) 2>&1 | grep -v amdgpu.ids > $O/r05_step_stamps_c5_share.txt
echo "[32-agent step forms]"
(echo "# bench.py --config c5 --envs E: single_step_launches (256 recorded steps replayed from a hipGraph, HIP events); MAPF_TUNE step_delta=2 forces the LDS delta-row form (with bitmaps), + bitmap_pairs=0 without them, step_delta=0 the plain step"; ENVS="16384 32768 65536 131072" FORMS="2 0" bash tools/exp/step32_ab.sh; echo "# the same with MAPF_TUNE=bitmap_pairs=0 in the environment is not expressible through step32_ab.sh's own MAPF_TUNE; run directly:"; for envs in 16384 65536 131072; do MAPF_TUNE=step_delta=2,bitmap_pairs=0 python3 bench.py --config c5 --envs $envs --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-scalar-env --no-policy-rollout --no-per-gpu-shapes 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); s=d['single_step_launches']; print('envs $envs MAPF_TUNE=step_delta=2,bitmap_pairs=0: %.3f us per launch, frac %.4f  %s' % (s['roofline']['ms_per_launch_hip_events']*1e3, s['roofline']['frac'], s['kernel'][:70]))"; done) > $O/r05_step32_forms.txt 2>&1
echo "[T sweep]"
python3 tools/rollout_T_sweep.py 2>&1 | grep -v amdgpu.ids > $O/r05_rollout_T_sweep.txt
(T_SWEEP=1,2,4,8 python3 tools/rollout_T_sweep.py c3 65536) 2>&1 | grep -v amdgpu.ids >> $O/r05_rollout_T_sweep.txt
echo "[transitions profiles]"
bash tools/exp/transitions_prof.sh 8 20000 compact 2>&1 | grep -v "^E2026\|^W2026\|amdgpu.ids\|at::native\|rocclr" > $O/r05_transitions_a8_q20000_compact.txt
bash tools/exp/transitions_prof.sh 4 2000000 compact 2>&1 | grep -v "^E2026\|^W2026\|amdgpu.ids\|at::native\|rocclr" > $O/r05_transitions_a4_q2M_compact.txt
echo "[bench configs]"
python3 tools/bench_configs.py 2>&1 | grep -v amdgpu.ids > $O/r05_bench_configs.txt
echo "[rehearsals]"
python3 bench.py --gpus 2 --dist-backend gloo --share-device --steps 20 --warmup 5 --no-side-legs --no-cpu-baseline > $O/r05_bench_2rank_gloo_rehearsal.json 2> $O/reh2.err
python3 bench.py --gpus 3 --dist-backend gloo --share-device --steps 5 --warmup 2 --repeats 1 --no-side-legs --no-cpu-baseline > $O/r05_bench_3rank_gloo_rehearsal.json 2> $O/reh3.err
python3 bench.py --gpus 1 --dist-backend nccl --force-dist --steps 20 --warmup 5 --no-side-legs --no-cpu-baseline > $O/r05_bench_rccl_world1_force_dist.json 2> $O/reh1.err
echo "[soak]"
(python3 tools/soak_parity.py 512 64; python3 tools/soak_parity.py 256 64 c5 16384) 2>&1 | grep -v amdgpu.ids > $O/r05_soak_parity.txt
echo done
