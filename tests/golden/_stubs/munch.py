"""Stand-in for munch.DefaultMunch(default, mapping): dict with attribute access, missing -> default."""
import copy


class DefaultMunch(dict):
    def __init__(self, default=None, mapping=None):
        super().__init__(mapping or {})
        object.__setattr__(self, '_dm_default', default)

    def __getattr__(self, key):
        if key.startswith('__'):
            raise AttributeError(key)
        return self.get(key, object.__getattribute__(self, '_dm_default'))

    def __setattr__(self, key, value):
        self[key] = value

    def __deepcopy__(self, memo):
        return DefaultMunch(object.__getattribute__(self, '_dm_default'),
                            {k: copy.deepcopy(v, memo) for k, v in self.items()})

    def __reduce__(self):
        return (DefaultMunch, (object.__getattribute__(self, '_dm_default'), dict(self)))
