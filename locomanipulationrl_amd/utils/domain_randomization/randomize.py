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
Anything else (scale, mass, density, material_properties, stiffness ...) raises NotImplementedError when
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
        dr_config = self._cfg.get("domain_randomization", None)
        if dr_config is not None:
            randomize = dr_config.get("randomize", False)
            randomization_params = dr_config.get("randomization_params", None)
            if randomize and randomization_params is not None:          # randomize.py:52-56
                self.randomize = True
                self.min_frequency = int(dr_config.get("min_frequency", 1))

    # ------------------------------------------------------------------ reference entry points
    def apply_on_startup_domain_randomization(self, task):
        """randomize.py:58-117: scale / mass / density on_startup.  A model-table change, not a per-step quantity: not supported."""
        if not self.randomize:
            return
        params = self._cfg["domain_randomization"]["randomization_params"]
        for group in ("rigid_prim_views", "articulation_views"):
            for view, attrs in (params.get(group) or {}).items():
                for attribute, entry in (attrs or {}).items():
                    if entry is not None and "on_startup" in entry:
                        raise NotImplementedError(f"domain randomisation of {group}.{view}.{attribute} on_startup is not implemented "
                                                  "(robot scale / mass / density change the compiled model table)")

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
