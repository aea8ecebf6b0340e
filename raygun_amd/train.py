"""`raygun-train conf.json` — mirror of raygun.train.train (reference: src/raygun/train.py:52-74): read the config, build
<framework>.systems.<system>(config_path), call system.train().  LSF submission (cluster_train, :15-49) and the folder
walker (batch_train, :77-150) are cluster tooling and out of scope.

Data parallelism has no reference counterpart (single GPU): when the process was started by
`python -m torch.distributed.run` (WORLD_SIZE > 1 in the environment) — or by `python -m raygun_amd.train --gpus N conf.json`,
which starts N such processes itself — every rank builds the same system, rank 0's weights are broadcast, and the gradients
are averaged over RCCL before each optimiser step (raygun_amd.parallel.GradSync)."""
import os
import sys

from .load_system import load_system


def train(config_path=None, batches=None):
    """Train system (CLI: python -m raygun_amd.train conf.json).  `batches`: optional iterable of batch dicts; by default the
    system's own batch source is used (BaseSystem.setup_datapipes: the config's zarr sources, or synthetic data)."""
    if config_path is None:
        config_path = sys.argv[1]
    if not isinstance(config_path, dict):
        config_path = os.path.realpath(config_path)
    from . import parallel
    rank, local, world = parallel.init_from_env()
    system = load_system(config_path)
    if world > 1:
        system.distributed = (rank, local, world)
    system.logger.info("System loaded. Training...")
    out = system.train(batches=batches)
    system.logger.info("Done training!")
    return out


def launch(n_gpus, argv, timeout=None):
    """Start one training process per GPU from a parent that never touches the GPU (parallel.launch_ranks: the ranks are
    supervised — the first failing rank, or the timeout, ends them all); returns the first non-zero exit code."""
    from .parallel import launch_ranks
    return launch_ranks([sys.executable, "-m", "raygun_amd.train"] + list(argv), n_gpus, timeout=timeout)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if argv and argv[0] == "--gpus":
        n = int(argv[1])
        sys.exit(launch(n, argv[2:]) if n > 1 else (train(argv[2]) and 0))
    train(argv[0])


if __name__ == "__main__":
    main()
