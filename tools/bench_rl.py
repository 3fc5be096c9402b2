#!/usr/bin/env python3
"""RL-style loop on device buffers (BASELINE config 5: a learner consuming the vectorised QRMSA env, N GPUs):
observation + action mask -> masked policy in PyTorch-ROCm -> step(actions), every launch on torch's current stream
(ongym_set_stream: no host synchronisation inside the loop).

    python tools/bench_rl.py [--gpus N] [--batch B] [--steps K] [--learner [--horizon H]]

Default: a masked random policy (env-side cost only).  `--learner`: a masked actor-critic MLP (368 -> 512 -> 512 -> 9601
logits + value head, bf16 autocast) sampled every step and updated with Adam on n-step returns every `--horizon` steps —
the data flow of the reference's MaskablePPO scripts (examples/ONDM_2025/train_multi_masked_ppo.py:410-458: 14
SubprocVecEnv workers feeding one learner; no PPO library is installed in this image).  The env writes into / reads from
torch device tensors (`io_device=1`, torch.Tensor.data_ptr()); nothing crosses PCIe inside the loop.

`--gpus N`: one process per GPU (spawned as a child `python -m torch.distributed.run` before anything touches the GPU, or
taken from the launcher's environment); rank k steps the global replicas shard_bounds(N*B, k, N) with request streams keyed
by the GLOBAL replica index, owns a copy of the learner and averages the gradients over the ranks with one bucketed RCCL
all-reduce per update (optical_networking_gym/_dist.py:allreduce_mean_gradients).  Rank 0 prints ONE JSON line: `value` =
global env-steps/s (max wall time over ranks), per-rank rates, the process group's size and backend.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "optical-networking-gym_amd"), REPO]


def spawn_ranks(gpus):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--batch", type=int, default=16384, help="replicas per GPU")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=300, help="first-fit steps that fill the network before the loop")
    ap.add_argument("--learner", action="store_true")
    ap.add_argument("--horizon", type=int, default=16)
    ap.add_argument("--torch-sampler", action="store_true", help="masked sampling with torch ops instead of ongym_sample_actions")
    ap.add_argument("--own-stream", action="store_true", help="round-2 behaviour: env on its own stream, three host syncs per step")
    args = ap.parse_args()
    if "RANK" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus))
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")

    import torch
    import bench
    from optical_networking_gym import _native as nat
    from optical_networking_gym._dist import (allreduce_mean_gradients, gather_per_rank, init_process_group,
                                              reduce_run_statistics, shard_bounds)
    from optical_networking_gym.envs.batched import BatchedQRMSAEnv
    rehearse = os.environ.get("ONGYM_BENCH_REHEARSE") == "1"       # ranks share the GPUs, gloo: plumbing check only
    device = local_rank % torch.cuda.device_count() if rehearse else local_rank
    torch.cuda.set_device(device)
    dist = init_process_group("gloo" if rehearse else "nccl", device) if (world > 1 or "RANK" in os.environ) else None
    red_dev = "cpu" if rehearse else "cuda"
    if rank == 0:
        import __graft_entry__ as entry
        entry.build()
    if dist:
        dist.barrier()

    B = args.batch
    base, _ = shard_bounds(B * world, rank, world)
    wl = bench.WORKLOADS["nsfnet320"]
    env = BatchedQRMSAEnv(tables=bench.build_tables(wl["topology"]), modulations=bench.jocn_modulations(), batch_size=B,
                          device=device, num_spectrum_resources=wl["S"], capacity=wl["capacity"], episode_length=1000,
                          auto_reset=True, load=wl["load"], bit_rate_selection="discrete", bit_rates=wl["bit_rates"],
                          io_device=True)
    c = env.holder.struct
    obs_dim, nact = 3 + c.k_paths + c.k_paths * c.n_mods_consider * 12, c.k_paths * c.n_mods_consider * c.n_slots + 1
    dev = torch.device("cuda", device)
    obs = torch.empty((B, obs_dim), dtype=torch.float32, device=dev)
    mask = torch.empty((B, nact), dtype=torch.uint8, device=dev)
    actions = torch.empty(B, dtype=torch.int32, device=dev)
    recs = torch.empty((B, nat.STEP_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    env.seed(1, replica_base=base)
    env.reset()
    env.step_policy(args.warmup, record=False)      # fill the network with first fit
    env.sync()
    shared = not args.own_stream
    if shared:
        env.set_stream(torch.cuda.current_stream().cuda_stream)

    def env_observe():
        env._check(env.lib.ongym_observe(env._h, obs.data_ptr(), mask.data_ptr()), "observe")
        if not shared:
            env.sync()

    def env_step():
        if not shared:
            torch.cuda.synchronize()
        env._check(env.lib.ongym_step_actions(env._h, actions.data_ptr(), recs.data_ptr()), "step")
        if not shared:
            env.sync()

    # masked uniform sampling in two levels (blocks of 64 actions): the mask is read once as bytes; torch.multinomial on
    # mask.float() moved 5x the mask's bytes and cost as much as the observation kernel
    assert (nact - 1) % 64 == 0, "k * M * S must be a multiple of 64 for the block view (S = 320: 9600)"
    nblk = (nact - 1) // 64
    rows = torch.arange(B, device=dev)

    def sample_masked():
        blocks = mask[:, :nact - 1].view(B, nblk, 64)                         # a view: nothing is copied
        bcum = blocks.sum(2, dtype=torch.int32).cumsum(1, dtype=torch.int32)  # valid actions up to and including each block
        total = bcum[:, -1] + 1                                               # + the reject action, always valid (:766)
        r = torch.minimum((torch.rand(B, device=dev) * total).to(torch.int32), total - 1)     # the r-th valid action (0-based)
        bi = torch.searchsorted(bcum, r.unsqueeze(1), right=True).squeeze(1)  # its block (nblk: the reject action)
        bic = bi.clamp(max=nblk - 1)
        before = torch.where(bi > 0, bcum.gather(1, (bi - 1).clamp(min=0).unsqueeze(1)).squeeze(1), torch.zeros_like(r))
        inner = torch.searchsorted(blocks[rows, bic].to(torch.int32).cumsum(1, dtype=torch.int32), (r - before).unsqueeze(1),
                                   right=True).squeeze(1)
        return torch.where(bi >= nblk, torch.full_like(r, nact - 1), (bic * 64 + inner).to(torch.int32))

    draw = [0]

    def rl_step():
        env_observe()
        if args.torch_sampler:
            actions.copy_(sample_masked())                                    # masked random policy, torch ops
        else:                                                                 # ... or the library's sampler (one small kernel)
            env._check(env.lib.ongym_sample_actions(env._h, mask.data_ptr(), 7, draw[0], actions.data_ptr()), "sample")
            draw[0] += 1
        env_step()

    if args.learner:
        torch.manual_seed(0)          # every rank starts from the same weights; averaged gradients keep them equal
        body = torch.nn.Sequential(torch.nn.Linear(obs_dim, 512), torch.nn.Tanh(), torch.nn.Linear(512, 512), torch.nn.Tanh()).to(dev)
        pi_head, v_head = torch.nn.Linear(512, nact).to(dev), torch.nn.Linear(512, 1).to(dev)
        params = list(body.parameters()) + list(pi_head.parameters()) + list(v_head.parameters())
        opt = torch.optim.Adam(params, lr=3e-4)
        r_off = nat.STEP_DTYPE.fields["reward"][1]
        logps, values, rewards = [], [], []

        def rl_step():   # noqa: F811
            env_observe()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                h = body(obs)
                logits, v = pi_head(h).float(), v_head(h).float().squeeze(1)
            logits = logits.masked_fill(mask == 0, -1e9)
            dist_ = torch.distributions.Categorical(logits=logits)
            a = dist_.sample()
            actions.copy_(a)
            env_step()
            logps.append(dist_.log_prob(a)); values.append(v)
            rewards.append(recs[:, r_off:r_off + 8].contiguous().view(torch.float64).squeeze(1).float())
            if len(rewards) == args.horizon:
                ret, rets = torch.zeros(B, device=dev), []
                for r in reversed(rewards):
                    ret = r + 0.99 * ret
                    rets.append(ret)
                rets = torch.stack(rets[::-1]); vs = torch.stack(values); lp = torch.stack(logps)
                adv = (rets - vs).detach()
                loss = -(lp * adv).mean() + 0.5 * (rets - vs).pow(2).mean()
                opt.zero_grad(set_to_none=True)
                loss.backward()
                allreduce_mean_gradients(params, dist if not rehearse else None)     # ONE bucketed all-reduce (RCCL)
                opt.step()
                logps.clear(); values.clear(); rewards.clear()

    for _ in range(5 if not args.learner else args.horizon):
        rl_step()

    def fence():
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        env.sync()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rl_step()
    fence()
    dt = time.perf_counter() - t0
    acc = float(recs.cpu().numpy().view(nat.STEP_DTYPE)["accepted"].mean())
    import numpy as np
    _, dt_max, _ = reduce_run_statistics(np.zeros(1), dt, 0.0, dist, device=red_dev)
    per_rank = gather_per_rank([B * args.steps / dt, acc], dist, device=red_dev)
    if rank == 0:
        what = (f"observe + actor-critic MLP forward/sample + step, Adam update (gradients averaged over {world} rank(s)) every "
                f"{args.horizon} steps" if args.learner else "observe + masked sampling in torch + step")
        print(json.dumps({
            "metric": "RL env-steps/s (BASELINE config 5: learner loop consuming the vectorised QRMSA env)",
            "value": B * world * args.steps / dt_max, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
            "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "data": "synthetic",
            "config": {"workload": f"QRMSA nsfnet320, gen_observation path: {what}", "batch_per_gpu": B,
                       "global_batch": B * world, "stream": "caller's (torch current stream)" if shared else "own + host syncs",
                       "learner": bool(args.learner), "horizon": args.horizon if args.learner else None},
            "process_group": {"world_size": dist.get_world_size() if dist else 1, "backend": dist.get_backend() if dist else None,
                              "per_rank_value": [r[0] for r in per_rank], "per_rank_accepted": [r[1] for r in per_rank]},
            "rehearsal": rehearse or None}), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
