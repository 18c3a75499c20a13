"""Drop-in shaped like the reference's `localization_python` package
(localization_python/localization_python/__init__.py): LocalizationNode and main."""
from . import localization_node, optimize_global_map_pose  # noqa: F401
from .localization_node import LocalizationNode, main  # noqa: F401
from .optimize_global_map_pose import MapBuilder, make_map_data  # noqa: F401
