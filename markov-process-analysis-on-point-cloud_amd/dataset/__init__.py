"""Readers for the on-disk formats of the reference's datasets: same class
names, constructor arguments and item layout as the reference's `dataset/` package; every
farthest-point sampling they do runs on the gfx950 FPS kernel, batched over shapes."""
