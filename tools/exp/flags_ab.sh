for i in 1 2; do
for lib in default variants/libmapf_hip_nomisched.so variants/libmapf_hip_nopost.so variants/libmapf_hip_relaxed.so; do
  if [ "$lib" = default ]; then unset MAPF_HIP_LIB; else export MAPF_HIP_LIB=$PWD/gym-mapf_amd/gym_mapf_amd/lib/$lib; fi
  for f in "" "--config c4 --envs 32768" "--config c5 --envs 16384" "--config c5"; do
    echo -n "[$lib] [$f] "
    python3 bench.py $f --steps 10 --warmup 3 --repeats 3 --no-side-legs --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f G  frac %.4f' % (d['value']/1e9, d['roofline']['frac']))"
  done
done
done
