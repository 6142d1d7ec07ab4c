"""Command line entry points (mirror of ``tc_gan/run``)."""
