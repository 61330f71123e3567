"""Overlay of the MI355X hot path onto the reference's in-tree `torchrl` package.

`install()` imports the reference package (it must be importable: /path/to/vision4leg on sys.path) and rebinds ONLY the
hot-path names — the classes SURVEY.md §8(b) lists — on the reference's own modules. Everything else the starters import
(`torchrl.algo.VMPO`, `torchrl.utils.Logger`, `torchrl.env.get_vec_env`, the other policies / algorithms / collectors)
stays the reference's, so `starter/ppo_*.py` runs unchanged:

    import vision4leg_amd.overlay as overlay
    overlay.install()                 # pf.explore / vf / PPO.update / GAE on the HIP engine, reference collector
    overlay.install(fast_path=True)   # + fused rollout step, HBM-resident replay buffer, pinned observation upload

Rebinding is by name on every already-imported `torchrl.<pkg>` module that defines the name (the package and the
sub-module that holds the class: `from torchrl.algo import PPO`, `from torchrl.algo.on_policy.ppo import PPO` and
`torchrl.networks.nets.LocoTransformer` all resolve to the HIP class afterwards).
"""
import importlib
import sys

# reference package -> hot-path names taken from the same-named shell package
HOT_NAMES = {
    "networks": ["MLPBase", "NatureEncoder", "NatureFuseEncoder", "LocoTransformerEncoder", "TransformerEncoder",
                 "Net", "ImpalaEncoderProjNet", "LocoTransformer", "NatureEncoderProjNet", "Transformer"],
    "policies": ["GaussianContPolicyBasicBias", "GaussianContPolicyImpalaEncoderProj",
                 "GaussianContPolicyLocoTransformer", "GaussianContPolicyNatureEncoderProj",
                 "GaussianContPolicyTransformer"],
    "algo": ["PPO"],
    "replay_buffers": ["OnPolicyReplayBuffer"],
}
# names the reference does not have (the MI355X-native additions); attached to the package for `from torchrl.x import y`
NEW_NAMES = {
    "policies": ["RolloutActor"],
    "replay_buffers": ["DeviceOnPolicyReplayBuffer"],
}
# fast_path=True: the starters' `OnPolicyReplayBuffer(...)` / `VecOnPolicyCollector(...)` construct these instead
FAST_NAMES = {
    "replay_buffers": {"OnPolicyReplayBuffer": "DeviceOnPolicyReplayBuffer"},
    "collector": {"VecOnPolicyCollector": "VecOnPolicyCollector"},
}

_installed = {}


def _rebind(ref_pkg_name, name, obj, always_on_package=False):
    """Set `name` on torchrl.<pkg> and on each imported sub-module of it that already defines `name`."""
    hits = []
    prefix = ref_pkg_name + "."
    for modname, mod in list(sys.modules.items()):
        if mod is None or not (modname == ref_pkg_name or modname.startswith(prefix)):
            continue
        if name in getattr(mod, "__dict__", {}) or (always_on_package and modname == ref_pkg_name):
            _installed.setdefault((modname, name), mod.__dict__.get(name))
            setattr(mod, name, obj)
            hits.append(modname)
    return hits


def install(fast_path=False, reference_package="torchrl"):
    """Rebind the hot-path classes on the reference's `torchrl`. Returns {"pkg.name": [modules patched]}."""
    ref = importlib.import_module(reference_package)
    if getattr(ref, "__v4l_shell__", False):
        raise RuntimeError("vision4leg_amd.overlay: `%s` resolves to the HIP shell itself; put the reference "
                           "checkout first on sys.path" % reference_package)
    report = {}
    for pkg, names in HOT_NAMES.items():
        ref_name = "%s.%s" % (reference_package, pkg)
        importlib.import_module(ref_name)
        shell = importlib.import_module("vision4leg_amd.torchrl." + pkg)
        for name in names:
            hits = _rebind(ref_name, name, getattr(shell, name))
            if not hits:
                raise RuntimeError("vision4leg_amd.overlay: the reference has no %s.%s to replace" % (ref_name, name))
            report["%s.%s" % (pkg, name)] = hits
    for pkg, names in NEW_NAMES.items():
        shell = importlib.import_module("vision4leg_amd.torchrl." + pkg)
        for name in names:
            report["%s.%s" % (pkg, name)] = _rebind("%s.%s" % (reference_package, pkg), name, getattr(shell, name), True)
    if fast_path:
        for pkg, names in FAST_NAMES.items():
            ref_name = "%s.%s" % (reference_package, pkg)
            importlib.import_module(ref_name)
            shell = importlib.import_module("vision4leg_amd.torchrl." + pkg)
            for name, shell_name in names.items():
                report["%s.%s" % (pkg, name)] = _rebind(ref_name, name, getattr(shell, shell_name))
    return report


def uninstall():
    """Put the reference's own objects back (tests)."""
    for (modname, name), old in list(_installed.items()):
        mod = sys.modules.get(modname)
        if mod is not None:
            if old is None:
                mod.__dict__.pop(name, None)
            else:
                setattr(mod, name, old)
    _installed.clear()
