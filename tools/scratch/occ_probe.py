import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, "optical-networking-gym_amd"), REPO]
import bench
from optical_networking_gym.envs.batched import BatchedQRMSAEnv
wl = bench.WORKLOADS["nobeleu768"]
tb = bench.build_tables(wl["topology"])
for cap in (704, 640, 576, 512, 448, 384):
    env = BatchedQRMSAEnv(tables=tb, modulations=bench.jocn_modulations(), batch_size=64, num_spectrum_resources=wl["S"], capacity=cap,
                          episode_length=1000, auto_reset=True, load=wl["load"], bit_rate_selection="discrete", bit_rates=wl["bit_rates"])
    env.seed(1); env.reset()
    print(cap, env.occupancy())
    env.close()
