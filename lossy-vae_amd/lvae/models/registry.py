"""Model registry: `get_model(name, *args, **kwargs)` (reference contract: lvae/models/registry.py:4-15).
Unknown names raise KeyError; re-registering a name prints a warning and overrides, as the reference does."""
_all_models = dict()

_YELLOW, _RESET = '[93m', '[0m'


def register_model(func):
    name = func.__name__
    if name in _all_models:
        print(f'{_YELLOW}Warning: model function *{name}* is multiply defined.{_RESET}')
    _all_models[name] = func
    return func


def get_model(name, *args, **kwargs):
    return _all_models[name](*args, **kwargs)
