"""``cattrs.strategies`` of the test stand-in."""


def configure_union_passthrough(union, converter):
    return None


def include_subclasses(*a, **k):
    return None
