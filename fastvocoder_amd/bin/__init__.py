"""Entry points with the reference's names (bin/__init__.py:1-5); only the
inference-side modes are built (train / preprocess are out of scope)."""
from .synthesize import run_synthesizer  # noqa: F401
from .test import run_test  # noqa: F401
from .publish import run_publisher  # noqa: F401
