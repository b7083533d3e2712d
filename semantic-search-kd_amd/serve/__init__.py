"""HTTP surface of the search path (FastAPI), wire-compatible with the reference's src/serve."""
from .app import AppState, app_state, create_app  # noqa: F401
