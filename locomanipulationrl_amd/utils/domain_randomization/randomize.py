"""Domain randomisation front end (reference: utils/domain_randomization/randomize.py:39-578).

The reference's `Randomizer` reads the `domain_randomization` block of the task YAML
(cfg/task/QuadrupedPoseControl.yaml:102-173), adds observation / action noise in Python (:212-306) and registers the physics
attributes with omni.replicator.isaac.  Here the same YAML block is compiled into the engine's parameter block
(`lm_params.dr[]`, include/lm_engine.h) and *everything is sampled inside the one `lm_step` launch* (kernel `k_step_dr`): action
noise before the clipActions clamp, per-env gravity / base-link force / max efforts / max joint velocities for the control step,
observation noise on `obs_buf` before the clipObservations clamp.  The class keeps the reference's attribute and method names so
that the task / wrapper code reads the same (`randomize`, `min_frequency`, `set_up_domain_randomization`,
`apply_on_startup_domain_randomization`, `apply_actions_randomization`, `apply_observations_randomization`).

Supported entries (everything the reference's YAMLs enable):
    observations / actions        on_reset + on_interval, additive | scaling, gaussian | uniform | loguniform
    simulation.gravity            on_interval or on_reset, additive | scaling | direct, per-component parameters
    rigid_prim_views.<base link>.force                 "
    articulation_views.<robot>.max_efforts             "   (scalar parameters, one draw per joint)
    articulation_views.<robot>.joint_max_velocities    "
    articulation_views.<robot>.damping                 "   (the viscous joint damping of the PD-actuator tasks; no effect on velocity-drive tasks,
                                                           whose joint_damping is 0)
    articulation_views.<robot>.joint_friction          accepted and ignored with a warning (joint friction itself is not modelled)
    articulation_views.<robot>.scale on_startup        accepted with a warning: factors drawn and recorded, not applied (see
                                                       apply_on_startup_domain_randomization: the reference's call does not rescale the links either)
Anything else (mass, density, material_properties, stiffness ...) raises NotImplementedError when
`randomize: True` - a silently ignored randomisation would be worse than a loud one."""
from __future__ import annotations

from typing import List

import numpy as np

from ...engine_config import (DR_ACT_INTERVAL, DR_ACT_RESET, DR_BASE_FORCE, DR_CHANNELS, DR_DISTRIBUTIONS, DR_GRAVITY, DR_JOINT_DAMPING, DR_MAX_EFFORT,
                              DR_MAX_VELOCITY, DR_OBS_INTERVAL, DR_OBS_RESET, DR_OPERATIONS, DRChannel)

_ON_RESET_KEYS = ("operation", "distribution", "distribution_parameters")
_ON_INTERVAL_KEYS = ("frequency_interval", "operation", "distribution", "distribution_parameters")


def _channel(where: str, entry: dict, trigger: str, vector: bool) -> DRChannel:
    need = _ON_INTERVAL_KEYS if trigger == "on_interval" else _ON_RESET_KEYS
    if not set(need).issubset(entry.keys()):          # randomize.py:182-190
        raise ValueError(f"Please ensure the following randomization parameters for {where} {trigger} are provided: " + ", ".join(need) + ".")
    op, dist = str(entry["operation"]), str(entry["distribution"])
    if op not in DR_OPERATIONS or dist not in DR_DISTRIBUTIONS:
        raise ValueError(f"{where} {trigger}: unsupported operation {op!r} or distribution {dist!r}")
    prm = np.asarray(entry["distribution_parameters"], dtype=np.float64)
    if vector:
        if prm.shape != (2, 3):
            raise ValueError(f"{where} {trigger}: distribution_parameters must be [[3 values], [3 values]]")
        p0, p1 = prm[0].tolist(), prm[1].tolist()
    else:
        if prm.shape != (2,):
            raise ValueError(f"{where} {trigger}: distribution_parameters must be [a, b]")
        p0, p1 = [float(prm[0])] * 3, [float(prm[1])] * 3
    interval = int(entry["frequency_interval"]) if trigger == "on_interval" else 0
    if trigger == "on_interval" and interval < 1:
        raise ValueError(f"{where}: frequency_interval must be >= 1")
    return DRChannel(enabled=1, operation=DR_OPERATIONS[op], distribution=DR_DISTRIBUTIONS[dist], interval=interval, p0=p0, p1=p1)


class Randomizer:
    def __init__(self, sim_config):
        self._cfg = sim_config.task_config
        self._config = sim_config.config
        self.randomize = False
        self.min_frequency = 1
        self.active_domain_randomizations = dict()
        self._channels: List[DRChannel] = [DRChannel() for _ in range(DR_CHANNELS)]
        self._observations_dr_params = None
        self._actions_dr_params = None
        self.startup_scales = dict()          # (group, view) -> per-env factors drawn by apply_on_startup_domain_randomization
        dr_config = self._cfg.get("domain_randomization", None)
        if dr_config is not None:
            randomize = dr_config.get("randomize", False)
            randomization_params = dr_config.get("randomization_params", None)
            if randomize and randomization_params is not None:          # randomize.py:52-56
                self.randomize = True
                self.min_frequency = int(dr_config.get("min_frequency", 1))

    # ------------------------------------------------------------------ reference entry points
    def apply_on_startup_domain_randomization(self, task):
        """randomize.py:58-117: scale / mass / density on_startup.

        `articulation_views.<robot>.scale` is the one on_startup entry the reference's YAMLs carry (cfg/task/QuadrupedPoseControl.yaml:167-172,
        uniform [0.98, 1.02]).  In the reference it calls `view.set_local_scales` on the articulation root, which its authors note does not
        scale the robot ("checked; but not scaling the entire robot?", :110 of that YAML): the link geometry, inertias and joint frames PhysX
        simulates are unchanged.  Here the entry is therefore ACCEPTED: the per-env factors are drawn as the reference draws them (one
        synchronised factor per env, torch generator seeded with the config seed, randomize.py:60,308-350) and kept in `startup_scales`
        for inspection, a warning says that they do not enter the dynamics, and the compiled model table stays shared by all envs.
        mass / density on_startup entries would change that table per env and are refused.
        PARITY UNPINNED: whether PhysX rescales anything on `set_local_scales` rests on that question-marked author comment (the same line
        sits in every task YAML of the reference, e.g. JointLocomanipulation.yaml:166), and the factors come from a private torch.Generator
        seeded with the config seed, not from the global stream the reference seeds with `torch.manual_seed` before
        `randomize_scale_on_startup` - `startup_scales` records this repo's draws, not the reference's.  The entry is accepted (with the
        warning) rather than refused because the reference's own YAML block has to load entry for entry (`test_reference_dr_block_loads_verbatim`);
        `sim.engine.refuse_startup_scale: true` turns the warning into a NotImplementedError for users who prefer the loud failure."""
        if not self.randomize:
            return
        import warnings
        import torch
        params = self._cfg["domain_randomization"]["randomization_params"]
        for group in ("rigid_prim_views", "articulation_views"):
            for view, attrs in (params.get(group) or {}).items():
                for attribute, entry in (attrs or {}).items():
                    if entry is None or "on_startup" not in entry:
                        continue
                    st = entry["on_startup"]
                    if not set(_ON_RESET_KEYS).issubset(st.keys()):          # randomize.py:75-77,104-106
                        raise ValueError(f"Please ensure the following randomization parameters for {view} {attribute} on_startup are provided: "
                                         "operation, distribution, distribution_parameters.")
                    if attribute == "scale" and bool((self._cfg.get("sim", {}).get("engine", {}) or {}).get("refuse_startup_scale", False)):
                        raise NotImplementedError(f"{group}.{view}.scale on_startup does not enter this engine's dynamics (sim.engine.refuse_startup_scale is set)")
                    if attribute != "scale":
                        raise NotImplementedError(f"domain randomisation of {group}.{view}.{attribute} on_startup is not implemented "
                                                  "(mass / density change the compiled model table per env)")
                    n = int(self._cfg["env"]["numEnvs"])
                    g = torch.Generator().manual_seed(int(self._config.get("seed", 42)))
                    lo, hi = (float(x) for x in st["distribution_parameters"])
                    dist = str(st["distribution"])
                    if dist == "uniform":
                        f = lo + (hi - lo) * torch.rand(n, generator=g)
                    elif dist in ("loguniform", "log_uniform"):
                        f = torch.exp(np.log(lo) + (np.log(hi) - np.log(lo)) * torch.rand(n, generator=g))
                    elif dist in ("gaussian", "normal"):
                        f = lo + hi * torch.randn(n, generator=g)
                    else:
                        raise ValueError(f"{view} scale on_startup: unsupported distribution {dist!r}")
                    op = str(st["operation"])
                    if op not in ("scaling", "additive", "direct"):
                        raise ValueError(f"{view} scale on_startup: unsupported operation {op!r}")
                    self.startup_scales[(group, view)] = (1.0 * f) if op in ("scaling", "direct") else (1.0 + f)
                    self.active_domain_randomizations[(group, view, attribute, "on_startup")] = np.array(st["distribution_parameters"])
                    warnings.warn(f"{group}.{view}.scale on_startup: accepted; the factors ({lo}..{hi}) are drawn and kept in "
                                  "Randomizer.startup_scales but do not enter the dynamics (in the reference set_local_scales on the articulation "
                                  "root does not rescale the simulated links either)")

    def set_up_domain_randomization(self, task):
        """randomize.py:125-166: walk the YAML block; here it fills the engine's channel table."""
        if not self.randomize:
            return
        params = self._cfg["domain_randomization"]["randomization_params"]
        for opt, body in params.items():
            if opt == "observations":
                self._set_up_noise(task, "observations", body, DR_OBS_RESET, DR_OBS_INTERVAL)
                task.randomize_observations = True
                self._observations_dr_params = body
            elif opt == "actions":
                self._set_up_noise(task, "actions", body, DR_ACT_RESET, DR_ACT_INTERVAL)
                task.randomize_actions = True
                self._actions_dr_params = body
            elif opt == "simulation":
                for attribute, entry in (body or {}).items():
                    if attribute != "gravity":
                        raise NotImplementedError(f"domain randomisation of simulation.{attribute} is not implemented")
                    self._set_up_attribute(("simulation", attribute), entry, DR_GRAVITY, vector=True)
            elif opt == "rigid_prim_views":
                for view, attrs in (body or {}).items():
                    for attribute, entry in (attrs or {}).items():
                        if attribute in ("scale", "mass", "density"):
                            continue          # on_startup entries, handled (refused) above
                        if attribute != "force":
                            raise NotImplementedError(f"domain randomisation of rigid_prim_views.{view}.{attribute} is not implemented")
                        self._set_up_attribute(("rigid_prim_views", view, attribute), entry, DR_BASE_FORCE, vector=True)
            elif opt == "articulation_views":
                for view, attrs in (body or {}).items():
                    for attribute, entry in (attrs or {}).items():
                        if attribute == "scale":
                            continue
                        if attribute == "joint_friction":          # the joint friction coefficient itself is not modelled: scaling it changes nothing
                            import warnings
                            warnings.warn(f"articulation_views.{view}.joint_friction: joint friction is not modelled by this engine; entry ignored")
                            continue
                        ch = {"max_efforts": DR_MAX_EFFORT, "joint_max_velocities": DR_MAX_VELOCITY, "damping": DR_JOINT_DAMPING}.get(attribute)
                        if ch is None:
                            raise NotImplementedError(f"domain randomisation of articulation_views.{view}.{attribute} is not implemented")
                        self._set_up_attribute(("articulation_views", view, attribute), entry, ch, vector=False)
            else:
                raise ValueError(f"unknown domain randomisation group {opt!r}")

    def _set_up_noise(self, task, kind, body, ch_reset, ch_interval):
        if body is None:
            raise ValueError(f"{kind.capitalize()} randomization parameters are not provided.")          # randomize.py:170-171
        if "on_reset" in body:
            self._channels[ch_reset] = _channel(kind, body["on_reset"], "on_reset", vector=False)
            self.active_domain_randomizations[(kind, "on_reset")] = np.array(body["on_reset"]["distribution_parameters"])
        if "on_interval" in body:
            self._channels[ch_interval] = _channel(kind, body["on_interval"], "on_interval", vector=False)
            self.active_domain_randomizations[(kind, "on_interval")] = np.array(body["on_interval"]["distribution_parameters"])
        for c in (ch_reset, ch_interval):
            if self._channels[c].enabled and self._channels[c].operation == DR_OPERATIONS["direct"]:
                raise ValueError(f"{kind}: operation must be additive or scaling")

    def _set_up_attribute(self, key, entry, ch, vector):
        if entry is None:
            raise ValueError(f"Randomization parameters for {'.'.join(key)} is not provided.")
        if "on_reset" in entry and "on_interval" in entry:
            raise NotImplementedError(f"{'.'.join(key)}: give either on_reset or on_interval, not both")
        for trigger in ("on_reset", "on_interval"):
            if trigger in entry:
                self._channels[ch] = _channel(".".join(key), entry[trigger], trigger, vector)
                self.active_domain_randomizations[key + (trigger,)] = np.array(entry[trigger]["distribution_parameters"])

    # ------------------------------------------------------------------ engine side
    def engine_dr(self) -> dict:
        """Fields of EngineParams describing the randomisation (all channels off when randomize is False)."""
        if not self.randomize:
            return dict(dr_enabled=0, dr_min_frequency=1, dr=[DRChannel() for _ in range(DR_CHANNELS)])
        return dict(dr_enabled=1, dr_min_frequency=int(self.min_frequency), dr=list(self._channels))

    # The wrapper calls these two exactly where the reference does (vec_env_rlgames.py:56-58,70-72).  The noise has already been /
    # will be applied inside lm_step with the reference's counter semantics (randomize.py:212-306), so they hand the tensor through.
    def apply_actions_randomization(self, actions, reset_buf):
        return actions

    def apply_observations_randomization(self, observations, reset_buf):
        return observations
