"""Mirror of raygun.torch (networks, models, losses, optimizers, systems, predict) on the HIP kernels."""
