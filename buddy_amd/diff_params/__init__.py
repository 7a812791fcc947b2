from .edm import EDM  # noqa: F401
