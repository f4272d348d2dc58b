from . import stats  # noqa: F401
