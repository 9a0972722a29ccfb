from hidenn_fem_amd.loss import EnergyLoss2D, l2_projection_loss, bar_energy_loss  # noqa: F401
