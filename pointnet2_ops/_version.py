__version__ = "3.0.0+slide_amd"
