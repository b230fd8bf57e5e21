#!/usr/bin/env python3
"""How often does the plate pass through the robot's body?  

The reference keeps colliders on the frame, link1-4 and the transmission (Design/Scripts/setup_collisions.py:3-10); this engine collides the four
foot spheres only and relies on the task's own resets (plate-frame corner / knee / base tests, quadruped_manipulate_plate.py:576-603)
firing before a body contact could matter.  This tool counts, per env-step of the manipulation robots, whether the plate's box collider
(0.5 x 0.5 x 0.008 m, Design/ObjectURDF/plate.urdf) intersects
    frame    the frame's box  +-(0.075, 0.1838, 0.0395) m about the base origin (mesh AABB)
    hull     a link hull: the 6.25 mm capsule about a link axis (link4: dof2 joint -> knee, link3: knee -> foot, link1: dof3 joint -> knee, link2: knee
             -> foot), the last tenth next to the foot sphere excluded (the modelled contact)
    axis     a link AXIS inside the plate's slab itself (the hull is through the plate by more than its radius)
under (i) fresh U(-1, 1) actions for 300 steps at 4096 envs (the full-size test's protocol) and (ii) the policy tools/train_ppo.py's recipe has
learned after --timesteps, sampled for 300 steps.  Intersections are counted on the state AFTER each step, terminal states included; the *_live
counts leave out the steps whose own reset flag is raised (the task's plate-frame tests caught those: the episode ends there in the reference too).

    python tools/plate_intersection.py [--timesteps 4800] [--out profiles/r04_plate_intersection.json]       (GPU box)
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import locomanipulationrl_amd as lm
from locomanipulationrl_amd.engine_config import MODE_MANI
from locomanipulationrl_amd.model.robot_model import load_model

FRAME_HALF = (0.075, 0.1838, 0.0395)
HULL_R = 0.00625


def quat_to_mat(q):      # (N, 4) wxyz -> (N, 3, 3)
    w, x, y, z = q.unbind(-1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)


class Geometry:
    """Batched forward kinematics of the limb bodies on the device (the host mirror RobotModel.fk, vectorised) + the sample points of the frame box."""

    def __init__(self, rm, dev):
        self.rm, self.dev = rm, dev
        t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32, device=dev)
        self.Rt, self.pt, self.axis = t(rm.Rt).reshape(-1, 3, 3), t(rm.pt), t(rm.axis)
        self.foot_off = t(rm.contact_off)
        g = torch.linspace(-1, 1, 7, device=dev); a, b = torch.meshgrid(g, g, indexing="ij"); a, b = a.reshape(-1), b.reshape(-1); one = torch.ones_like(a)
        h = torch.tensor(FRAME_HALF, device=dev)
        faces = [torch.stack(p, -1) for s in (-1.0, 1.0) for p in ((s * one, a, b), (a, s * one, b), (a, b, s * one))]
        self.frame_pts = torch.cat(faces) * h                                           # (294, 3) on the frame box's surface, base frame
        gp = torch.linspace(-0.25, 0.25, 21, device=dev); pa, pb = torch.meshgrid(gp, gp, indexing="ij")
        self.plate_pts = torch.cat([torch.stack([pa.reshape(-1), pb.reshape(-1), torch.full((441,), z, device=dev)], -1) for z in (0.0, 0.008)])      # plate frame

    def body_poses(self, q12, R0, p0):
        rm = self.rm; N = q12.shape[0]
        qt = torch.zeros(N, 20, device=self.dev); qt[:, :12] = q12
        for c in range(8):
            D = q12[:, int(rm.clos_a[c])] - q12[:, int(rm.clos_b[c])]
            qt[:, int(rm.clos_p[c])] = float(rm.clos_s[c]) * 2 * torch.atan2(np.sqrt(2.0) * torch.sin(D / 2), torch.cos(D / 2))
        R, p = [R0], [p0]
        for k in range(1, rm.nb):
            par = int(rm.parent[k]); a = self.axis[k]; th = qt[:, int(rm.dof[k])]
            K = torch.tensor([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]], device=self.dev)
            Rq = torch.eye(3, device=self.dev) + torch.sin(th)[:, None, None] * K + (1 - torch.cos(th))[:, None, None] * (K @ K)
            R.append(R[par] @ self.Rt[k] @ Rq); p.append(p[par] + (R[par] @ self.pt[k]))
        return R, p

    def flags(self, state, ep, sl):
        """(frame, hull, axis, hull-at-the-rim) boolean vectors over the envs of slice `sl` (a manipulation block with parameters `ep`)."""
        s = state[:, sl].T
        N = s.shape[0]
        Rb = quat_to_mat(torch.tensor(ep.fixed_base_quat, device=self.dev).expand(N, 4)); pb = torch.tensor(ep.fixed_base_pos, device=self.dev).expand(N, 3)
        Rp = quat_to_mat(s[:, 40:44]); pp = s[:, 37:40]
        to_plate = lambda x: torch.einsum("nji,nkj->nki", Rp, x - pp[:, None, :])          # world (N, K, 3) -> plate frame
        to_base = lambda x: torch.einsum("nji,nkj->nki", Rb, x - pb[:, None, :])

        def in_slab(y, grow):      # the plate's box collider: |x|, |y| <= 0.25, z in [0, 0.008], grown by `grow`
            return (y[..., 0].abs() <= 0.25 + grow) & (y[..., 1].abs() <= 0.25 + grow) & (y[..., 2] >= -grow) & (y[..., 2] <= 0.008 + grow)
        fw = pb[:, None, :] + torch.einsum("nij,kj->nki", Rb, self.frame_pts)
        h = torch.tensor(FRAME_HALF, device=self.dev)
        pw = pp[:, None, :] + torch.einsum("nij,kj->nki", Rp, self.plate_pts)
        frame = in_slab(to_plate(fw), 0.0).any(1) | (to_base(pw).abs() <= h).all(-1).any(1)
        R, p = self.body_poses(s[:, 13:25], Rb, pb)
        rm = self.rm; segs = []
        for l in range(4):
            b = [int(x) for x in rm.limb_body_index[l]]          # shell, link4, link3, link1, link2
            cb = int(rm.contact_body[l]); foot = p[cb] + torch.einsum("nij,j->ni", R[cb], self.foot_off[l])
            segs += [(p[b[1]], p[b[2]], 1.0), (p[b[2]], foot, 0.9), (p[b[3]], p[b[4]], 1.0), (p[b[4]], foot, 0.9)]
        tt = torch.linspace(0, 1, 11, device=self.dev)
        pts = torch.cat([(a[:, None, :] + (tt * f)[None, :, None] * (b_ - a)[:, None, :]) for a, b_, f in segs], 1)      # (N, 16 x 11, 3)
        y = to_plate(pts)
        hit = in_slab(y, HULL_R)
        # where a hull overlap sits: within 10 mm of the plate's rim (a foot has slipped off the edge and the link crosses the rim) or over the face
        rim = (hit & ((y[..., 0].abs() > 0.24) | (y[..., 1].abs() > 0.24))).any(1)
        return frame, hit.any(1), in_slab(y, 0.0).any(1), rim


def mani_blocks(task):
    eps = task.engine_params(); N = task.num_envs; split = task.split_env()
    if len(eps) == 1:
        return [(eps[0], slice(0, N))] if eps[0].mode == MODE_MANI else []
    return [(ep, sl) for ep, sl in ((eps[0], slice(0, split)), (eps[1], slice(split, N))) if ep.mode == MODE_MANI]


def count(env, geo, actions_fn, steps):
    task = env._task; blocks = mani_blocks(task); tot = dict(env_steps=0, frame=0, hull=0, axis=0, any=0, resets=0, frame_live=0, hull_live=0, any_live=0, hull_live_at_rim=0)
    # episode level: an episode is TOUCHED once a link hull overlaps the plate on a live step; its outcome is read when its reset flag rises
    # (goal_reset_buf = the reference's success flag, quadruped_manipulate_plate.py:617-631)
    epi = dict(episodes=0, touched=0, success=0, success_touched=0, steps_after_first_touch=0)
    N = task.num_envs; touched = torch.zeros(N, dtype=torch.bool, device=geo.dev); since = torch.zeros(N, dtype=torch.int64, device=geo.dev)
    obs = env.reset()["obs"]
    for t in range(steps):
        o, rew, done, _ = env.step(actions_fn(obs)); obs = o["obs"]
        st = task.engine.state
        for ep, sl in blocks:
            f, hl, ax, rim = geo.flags(st, ep, sl)
            tot["env_steps"] += int(f.numel()); tot["frame"] += int(f.sum()); tot["hull"] += int(hl.sum()); tot["axis"] += int(ax.sum()); tot["any"] += int((f | hl).sum())
            tot["resets"] += int(done[sl].sum())
            live = done[sl] == 0          # the step did not end the episode: an intersection here is one the task's own tests have NOT caught
            tot["frame_live"] += int((f & live).sum()); tot["hull_live"] += int((hl & live).sum()); tot["any_live"] += int(((f | hl) & live).sum()); tot["hull_live_at_rim"] += int((rim & live).sum())
            since[sl] += touched[sl].long(); touched[sl] |= hl & live
            ended = done[sl] != 0; good = ended & (task.goal_reset_buf[sl] != 0)
            epi["episodes"] += int(ended.sum()); epi["touched"] += int((ended & touched[sl]).sum()); epi["success"] += int(good.sum())
            epi["success_touched"] += int((good & touched[sl]).sum()); epi["steps_after_first_touch"] += int(since[sl][ended & touched[sl]].sum())
            touched[sl] &= ~ended; since[sl] *= (~ended).long()
    for k in ("frame", "hull", "axis", "any", "frame_live", "hull_live", "any_live", "hull_live_at_rim"):
        tot[k + "_pct"] = round(100.0 * tot[k] / max(tot["env_steps"], 1), 4)
    u = epi["episodes"] - epi["touched"]
    epi.update(touched_pct=round(100.0 * epi["touched"] / max(epi["episodes"], 1), 3), success_rate_untouched=round((epi["success"] - epi["success_touched"]) / max(u, 1), 4),
               success_rate_touched=round(epi["success_touched"] / max(epi["touched"], 1), 4), mean_steps_after_first_touch=round(epi["steps_after_first_touch"] / max(epi["touched"], 1), 2),
               # if every touched episode that failed had succeeded instead (the most a link collider could change)
               success_rate_all=round(epi["success"] / max(epi["episodes"], 1), 4),
               success_rate_upper_bound_with_colliders=round((epi["success"] + epi["touched"] - epi["success_touched"]) / max(epi["episodes"], 1), 4))
    tot["episodes"] = epi
    return tot


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--timesteps", type=int, default=4800); ap.add_argument("--num-envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=300); ap.add_argument("--out", default=""); ap.add_argument("--tasks", default="QuadrupedManipulatePlate,JointLocomanipulation,QuadrupedManipulatePlateCustomController")
    a = ap.parse_args(); dev = "cuda:0"; doc = {"source": "tools/plate_intersection.py", "envs": a.num_envs, "steps": a.steps, "ppo_timesteps": a.timesteps, "tasks": {}}
    from locomanipulationrl_amd.policies.mlp_model import SharedMLP
    from locomanipulationrl_amd.train.ppo import PPO
    for name in a.tasks.split(","):
        env = lm.make_env(name, num_envs=a.num_envs, seed=42)
        geo = Geometry(load_model(env._task.model_asset), dev)
        g = torch.Generator(device=dev).manual_seed(42)
        row = {"random_actions": count(env, geo, lambda obs: torch.rand(a.num_envs, 12, device=dev, generator=g) * 2 - 1, a.steps)}
        print(json.dumps({name: row}), flush=True)
        torch.manual_seed(42)
        model = SharedMLP(num_observations=env.observation_space.shape[0]).to(dev)
        ppo = PPO(env, model)
        hist = ppo.train(a.timesteps, log_every=1000, log=lambda r: None)
        ppo.rollout = None                      # evaluate step by step through VecEnvRLGames.step

        def act(obs):
            mean, log_std, _ = ppo._policy(obs)
            return mean + log_std.exp() * torch.randn_like(mean)
        row["trained_policy"] = count(env, geo, act, a.steps)
        row["trained_policy"]["success_rate_at_end_of_training"] = hist[-1].get("success_rate") if hist else None
        print(json.dumps({name: row}), flush=True)
        doc["tasks"][name] = row
        env.close()
    if a.out:
        json.dump(doc, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
