__version__ = "2.1.0-stub"
