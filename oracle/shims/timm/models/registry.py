"""Oracle shim (TEST INFRASTRUCTURE ONLY): timm.models.registry.register_model is a
pass-through decorator here."""


def register_model(fn):
    return fn
