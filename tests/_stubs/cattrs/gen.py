"""``cattrs.gen`` of the test stand-in: hook generators return functions that refuse to run (no serialisation here)."""


def _refuse(*a, **k):
    raise NotImplementedError("serialisation is outside the cattrs stand-in of the test suite")


def make_dict_unstructure_fn(cl, converter, **kw):
    return _refuse


def make_dict_structure_fn(cl, converter, **kw):
    return _refuse
