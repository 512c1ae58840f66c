"""Loss / metric / data helpers with the reference's module names (utils/dice_score.py, utils/boundary_loss.py,
utils/connected_component_loss.py, utils/post_process.py, utils/data_loading.py), backed by the HIP kernels in csrc/."""
