"""Consumer side of the OAKE feature files: access layers + ``LoadCLIPFeatures``."""
from .features import LoadCLIPFeatures, PackAccessLayer, PthAccessLayer, build_access_layer, pack

__all__ = ['LoadCLIPFeatures', 'PackAccessLayer', 'PthAccessLayer', 'build_access_layer', 'pack']
