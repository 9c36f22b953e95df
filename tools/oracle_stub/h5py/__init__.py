"""Empty stand-in: graph_trajectory.py imports h5py at module level; not used when unpickling."""
