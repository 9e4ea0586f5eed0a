"""Record the registered (env cfg, train cfg) value trees of the reference's tasks as data.

BUILD-CONTAINER ONLY (imports /root/reference through ref_loader's stubs).  For every task name given (default: the tasks this
package registers), `class_to_dict` of the instances in the reference's `task_registry` (`legged_gym/envs/__init__.py:114-198`,
`utils/helpers.py:43-58`) goes to tests/golden/task_configs.json.  tests/test_task_configs.py holds our config classes to it.

Usage:  python tools/refgen/make_task_config_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402

REPO = ref_loader.REPO_ROOT
sys.path.insert(0, REPO)


def clean(x):
    if isinstance(x, dict):
        return {k: clean(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [clean(v) for v in x]
    if isinstance(x, (int, float, str, bool)) or x is None:
        return x
    return repr(x)


def main():
    ref_loader.load_reference()
    import legged_gym.envs  # noqa: F401  (registers the tasks; first, as the reference's own scripts import it)
    from legged_gym.utils.helpers import class_to_dict
    from legged_gym.utils.task_registry import task_registry
    from extended_legged_gym_amd.envs import task_registry as ours
    names = sys.argv[1:] or sorted(ours.task_classes.keys())
    out = {}
    for n in names:
        env_cfg, train_cfg = task_registry.env_cfgs[n], task_registry.train_cfgs[n]
        out[n] = dict(task_class=task_registry.task_classes[n].__name__, env=clean(class_to_dict(env_cfg)), train=clean(class_to_dict(train_cfg)))
    path = os.path.join(REPO, "tests", "golden", "task_configs.json")
    json.dump(out, open(path, "w"), indent=0, sort_keys=True)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;", len(out), "tasks")


if __name__ == "__main__":
    main()
