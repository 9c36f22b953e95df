"""Name-only stand-in: the reference imports `SparseTensor` for isinstance checks."""


class SparseTensor:  # never instantiated on the hot path
    pass
