"""``cattrs.dispatch`` of the test stand-in (type aliases only)."""
from typing import Any, Callable

UnstructureHook = Callable[[Any], Any]
StructureHook = Callable[[Any, Any], Any]
