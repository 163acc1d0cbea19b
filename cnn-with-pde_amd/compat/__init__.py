"""Per-script aliases: ``from cnn_with_pde_amd.compat.mnist_test import DiffusionLayer`` gives
the drop-in for the class of the same name in the reference's ``mnist_test.py`` (and so on), so
a training script only changes the line that defines the layer."""
