"""Drop-in shaped like the reference's `localization_python` package
(localization_python/localization_python/__init__.py): LocalizationNode and main."""
from .localization_node import LocalizationNode, main  # noqa: F401
