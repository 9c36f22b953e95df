"""Stand-in for termcolor (graph_trajectory.py:?): colored() returns the string unchanged."""


def colored(s, *a, **k):
    return s
