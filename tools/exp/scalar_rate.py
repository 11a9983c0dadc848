import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, 'gym-mapf_amd'), ROOT]
import bench
for _ in range(3):
    print(bench.scalar_env_rate(2.0)['value'])
import cProfile, pstats, random
from gym_mapf_amd.envs.utils import create_mapf_env
from gym_mapf_amd.envs.vec_env import OptimizationCriteria
env = create_mapf_env('empty-8-8', 1, 2, 0.0, -1000.0, 100.0, -1.0, OptimizationCriteria.Makespan)
rng = random.Random(0); acts = [rng.randrange(env.nA) for _ in range(20000)]
def run():
    for a in acts:
        if env.step(a)[2]: env.reset()
run()
cProfile.run('run()', '/tmp/prof.out')
pstats.Stats('/tmp/prof.out').sort_stats('cumtime').print_stats(12)
