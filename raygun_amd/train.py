"""`raygun-train conf.json` — mirror of raygun.train.train (reference: src/raygun/train.py:52-74): read the config, build
<framework>.systems.<system>(config_path), call system.train().  LSF submission (cluster_train, :15-49) and the folder
walker (batch_train, :77-150) are cluster tooling and out of scope.

Data parallelism has no reference counterpart (single GPU): when the process was started by
`python -m torch.distributed.run` (WORLD_SIZE > 1 in the environment) — or by `python -m raygun_amd.train --gpus N conf.json`,
which starts N such processes itself — every rank builds the same system, rank 0's weights are broadcast, and the gradients
are averaged over RCCL before each optimiser step (raygun_amd.parallel.GradSync)."""
import os
import subprocess
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


def launch(n_gpus, argv):
    """Start one training process per GPU (torchrun-style environment, rendezvous on 127.0.0.1) from a parent that never
    touches the GPU; returns the first non-zero exit code."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(n_gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, "-m", "raygun_amd.train"] + list(argv), env=env))
    rc = 0
    for p in procs:
        rc = rc or p.wait()
    return rc


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if argv and argv[0] == "--gpus":
        n = int(argv[1])
        sys.exit(launch(n, argv[2:]) if n > 1 else (train(argv[2]) and 0))
    train(argv[0])


if __name__ == "__main__":
    main()
