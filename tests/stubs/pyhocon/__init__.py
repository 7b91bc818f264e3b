"""TEST INFRASTRUCTURE: a reader for the HOCON subset the reference's confs use (reconstruction/confs/*.conf) with the slice of pyhocon's API
the reference's Runner calls (exp_runner_generic_blender_val.py:43-129): ConfigFactory.parse_file, conf['a.b.c'] (read and write),
conf.get_int / get_float / get_bool / get_string / get_list with default=, and ``**conf['model.x']``.  pyhocon itself is not installed in this image.

Grammar handled: ``key = value`` / ``key : value`` / ``key { ... }``, nested objects, lists ``[a, b\\n c]`` (comma or newline separated),
``#`` and ``//`` comments, trailing commas, quoted and unquoted strings (paths like exp/lod0, ../), ints, floats, true/false/True/False."""
import re

__version__ = "0.0-o2345-test-stub"


class ConfigMissingException(KeyError):
    pass


class ConfigTree(dict):
    def _walk(self, key, create=False):
        node = self
        parts = key.split(".")
        for p in parts[:-1]:
            if p not in node or not isinstance(dict.__getitem__(node, p), dict):
                if not create:
                    raise ConfigMissingException(key)
                dict.__setitem__(node, p, ConfigTree())
            node = dict.__getitem__(node, p)
        return node, parts[-1]

    def __getitem__(self, key):
        node, last = self._walk(key)
        if not dict.__contains__(node, last):
            raise ConfigMissingException(key)
        return dict.__getitem__(node, last)

    def __setitem__(self, key, value):
        node, last = self._walk(key, create=True)
        dict.__setitem__(node, last, value)

    def __contains__(self, key):
        try:
            self[key]
            return True
        except KeyError:
            return False

    _MISSING = object()

    def get(self, key, default=_MISSING):
        try:
            return self[key]
        except KeyError:
            if default is ConfigTree._MISSING:
                raise
            return default

    def get_int(self, key, default=_MISSING):
        v = self.get(key, default)
        return v if v is default else int(v)

    def get_float(self, key, default=_MISSING):
        v = self.get(key, default)
        return v if v is default else float(v)

    def get_bool(self, key, default=_MISSING):
        v = self.get(key, default)
        if v is default:
            return v
        return v.lower() in ("true", "yes", "on") if isinstance(v, str) else bool(v)

    def get_string(self, key, default=_MISSING):
        v = self.get(key, default)
        return v if v is default else str(v)

    def get_list(self, key, default=_MISSING):
        v = self.get(key, default)
        return v if v is default else list(v)

    def get_config(self, key, default=_MISSING):
        return self.get(key, default)


_TOKEN = re.compile(r'''\s*(?:(?P<nl>\n)|(?P<punct>[{}\[\],=:])|"(?P<q>(?:[^"\\]|\\.)*)"|(?P<w>[^\s{}\[\],=:"#]+))''')


def _tokens(text):
    text = re.sub(r"(#|//)[^\n]*", "", text)
    pos, out = 0, []
    while pos < len(text):
        m = _TOKEN.match(text, pos)
        if not m:
            if text[pos:].strip() == "":
                break
            raise ValueError(f"pyhocon stub: cannot tokenise at {text[pos:pos + 40]!r}")
        pos = m.end()
        if m.group("nl"):
            out.append(("nl", "\n"))
        elif m.group("punct"):
            out.append(("p", m.group("punct")))
        elif m.group("q") is not None:
            out.append(("s", m.group("q")))
        else:
            out.append(("w", m.group("w")))
    return out


def _scalar(kind, v):
    if kind == "s":
        return v
    if v in ("true", "True"):
        return True
    if v in ("false", "False"):
        return False
    if v in ("null", "None"):
        return None
    for cast in (int, float):
        try:
            return cast(v)
        except ValueError:
            pass
    return v


class _Parser:
    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self):
        return self.t[self.i] if self.i < len(self.t) else (None, None)

    def skip(self, kinds=("nl",), puncts=(",",)):
        while self.i < len(self.t) and (self.t[self.i][0] in kinds or (self.t[self.i][0] == "p" and self.t[self.i][1] in puncts)):
            self.i += 1

    def obj(self, top=False):
        tree = ConfigTree()
        while True:
            self.skip()
            k, v = self.peek()
            if k is None:
                if not top:
                    raise ValueError("pyhocon stub: unterminated object")
                return tree
            if k == "p" and v == "}":
                self.i += 1
                return tree
            if k not in ("w", "s"):
                raise ValueError(f"pyhocon stub: key expected, got {v!r}")
            self.i += 1
            key = v
            k2, v2 = self.peek()
            if k2 == "p" and v2 in "=:":
                self.i += 1
                while self.peek()[0] == "nl":
                    self.i += 1
                tree[key] = self.value()
            elif k2 == "p" and v2 == "{":
                self.i += 1
                sub = self.obj()
                if key in tree and isinstance(tree[key], dict):
                    tree[key].update(sub)
                else:
                    tree[key] = sub
            else:
                raise ValueError(f"pyhocon stub: '=' or '{{' expected after {key!r}")

    def value(self):
        k, v = self.peek()
        if k == "p" and v == "{":
            self.i += 1
            return self.obj()
        if k == "p" and v == "[":
            self.i += 1
            out = []
            while True:
                self.skip()
                k, v = self.peek()
                if k == "p" and v == "]":
                    self.i += 1
                    return out
                out.append(self.value())
        if k in ("w", "s"):
            self.i += 1
            # unquoted strings may continue with further words on the same line ("a b"): the reference's confs never do
            return _scalar(k, v)
        raise ValueError(f"pyhocon stub: value expected, got {v!r}")


class ConfigFactory:
    @staticmethod
    def parse_string(text):
        return _Parser(_tokens(text)).obj(top=True)

    @staticmethod
    def parse_file(path):
        with open(path) as f:
            return ConfigFactory.parse_string(f.read())
