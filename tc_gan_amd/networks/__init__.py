"""BPTT conditional-WGAN on the GPU: host-side mirror of ``tc_gan/networks``."""
