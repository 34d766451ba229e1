#!/usr/bin/env python3
"""wgsl2cpp.py — mechanical WGSL -> C++ translation.  TEST INFRASTRUCTURE ONLY (oracle/): never part of the product.

Purpose: execute the REFERENCE'S OWN shader text on the CPU.  The arithmetic of bevy_terrain's preprocessing and
tiling prepass lives in WGSL files under /root/reference/src/shaders; the reference cannot be built here (no Rust,
no wgpu), so oracle/bt_oracle.c restates it by hand.  This tool removes the "same author read it twice" risk: it
reads the UNMODIFIED .wgsl files where they lie, resolves naga_oil's module syntax (#define_import_path, #import,
#ifdef / #else / #endif shader defs, virtual / override functions) and emits one C++ struct per compute pipeline whose
member functions are the shader's functions, statement for statement.  The C++ type system (wgsl_rt.hpp) then plays
WGSL's: abstract-int / abstract-float literals (AI / AF) that concretise on contact, vecN<T> / matCxR<T> with
component-wise operators, value semantics for structs and arrays.  What wgpu supplies at run time — bindings,
textureLoad / textureSampleLevel / textureGather, pack*unorm, atomics, the dispatch loop — is supplied by
ref_harness.cpp, nothing else.

Nothing is interpreted or special-cased per shader: the translator knows WGSL syntax, not bevy_terrain.
Outputs go to oracle/_ref/ (git-ignored); reference sources are never copied into the repository.

usage: wgsl2cpp.py <out.inc> <StructName> <defs,comma,separated|-> <entry.wgsl[,entry2.wgsl...]> <search dir>...
"""
import os
import re
import sys

# --------------------------------------------------------------------------------------------------------------
# naga_oil layer: shader defs, import paths
# --------------------------------------------------------------------------------------------------------------


def preprocess(text, defs):
    """#ifdef / #ifndef / #else / #endif with the given shader defs; inactive lines become empty (line numbers stay)."""
    out, stack, active = [], [], True
    for line in text.split("\n"):
        s = line.strip()
        m = re.match(r"#(ifdef|ifndef)\s+(\w+)", s)
        if m:
            cond = (m.group(2) in defs) == (m.group(1) == "ifdef")
            stack.append((active, cond))
            active = active and cond
            out.append("")
        elif s.startswith("#else"):
            parent, cond = stack[-1]
            active = parent and not cond
            out.append("")
        elif s.startswith("#endif"):
            active, _ = stack.pop()
            out.append("")
        elif s.startswith("#if"):
            raise SyntaxError("unsupported directive: " + s)
        else:
            out.append(line if active else "")
    if stack:
        raise SyntaxError("unterminated #ifdef")
    return "\n".join(out)


class Module:
    def __init__(self, path, filename, text):
        self.path, self.filename = path, filename
        self.imports = {}  # local name -> (module path, item name)
        self.items = {}  # name -> Item
        self.tag = re.sub(r"\W", "_", path.split("::")[-1])
        body = []
        for line in text.split("\n"):
            s = line.strip()
            if s.startswith("#define_import_path"):
                self.path = s.split()[1].rstrip(";")
                self.tag = re.sub(r"\W", "_", self.path.split("::")[-1])
                body.append("")
            elif s.startswith("#import"):
                m = re.match(r"#import\s+([\w:]+?)(?:::\{([^}]*)\})?\s*;?\s*$", s)
                if not m:
                    raise SyntaxError("bad import: " + s)
                if m.group(2) is not None:
                    for name in m.group(2).split(","):
                        name = name.strip()
                        if name:
                            self.imports[name] = (m.group(1), name)
                else:
                    mod, _, name = m.group(1).rpartition("::")
                    self.imports[name] = (mod, name)
                body.append("")
            else:
                body.append(line)
        self.text = "\n".join(body)


# --------------------------------------------------------------------------------------------------------------
# lexer
# --------------------------------------------------------------------------------------------------------------

TOKEN_RE = re.compile(
    r"(?P<ws>\s+)|(?P<comment>//[^\n]*|/\*.*?\*/)"
    r"|(?P<num>0[xX][0-9a-fA-F]+[iu]?|(?:\d+\.\d*|\.\d+|\d+)(?:[eE][+-]?\d+)?[fhiu]?)"
    r"|(?P<id>[A-Za-z_]\w*)"
    r"|(?P<op><<=|>>=|->|==|!=|<=|>=|&&|\|\||<<|>>|\+=|-=|\*=|/=|%=|&=|\|=|\^=|\+\+|--|::|[-+*/%&|^~!<>=.,;:(){}\[\]@])",
    re.S,
)


class Tok:
    __slots__ = ("kind", "text", "line")

    def __init__(self, kind, text, line):
        self.kind, self.text, self.line = kind, text, line

    def __repr__(self):
        return "%s:%r@%d" % (self.kind, self.text, self.line)


def lex(text):
    toks, pos, line = [], 0, 1
    while pos < len(text):
        m = TOKEN_RE.match(text, pos)
        if not m:
            raise SyntaxError("cannot lex at line %d: %r" % (line, text[pos : pos + 20]))
        kind = m.lastgroup
        if kind not in ("ws", "comment"):
            toks.append(Tok(kind, m.group(), line))
        line += m.group().count("\n")
        pos = m.end()
    toks.append(Tok("eof", "", line))
    return toks


# --------------------------------------------------------------------------------------------------------------
# parser (the WGSL subset these shaders use; anything else raises)
# --------------------------------------------------------------------------------------------------------------

TEMPLATED = {"vec2", "vec3", "vec4", "array", "atomic", "ptr", "bitcast", "texture_2d", "texture_2d_array", "texture_storage_2d",
             "mat2x2", "mat2x3", "mat2x4", "mat3x2", "mat3x3", "mat3x4", "mat4x2", "mat4x3", "mat4x4"}


class Node:
    def __init__(self, kind, **kw):
        self.kind = kind
        self.__dict__.update(kw)

    def __repr__(self):
        return "Node(%s)" % ", ".join("%s=%r" % kv for kv in self.__dict__.items())


class Parser:
    def __init__(self, toks, filename):
        self.toks, self.i, self.filename = toks, 0, filename

    # -- helpers
    def peek(self, k=0):
        return self.toks[self.i + k]

    def at(self, text):
        return self.peek().text == text and self.peek().kind in ("op", "id")

    def accept(self, text):
        if self.at(text):
            self.i += 1
            return True
        return False

    def expect(self, text):
        if not self.accept(text):
            t = self.peek()
            raise SyntaxError("%s:%d: expected %r, found %r" % (self.filename, t.line, text, t.text))

    def ident(self):
        t = self.peek()
        if t.kind != "id":
            raise SyntaxError("%s:%d: expected identifier, found %r" % (self.filename, t.line, t.text))
        self.i += 1
        return t.text

    def close_template(self):
        t = self.peek()
        if t.text == ">":
            self.i += 1
        elif t.text in (">>", ">=", ">>="):  # split the token: array<vec2<u32>>
            t.text = t.text[1:]
        else:
            raise SyntaxError("%s:%d: expected '>', found %r" % (self.filename, t.line, t.text))

    # -- attributes
    def attributes(self):
        attrs = []
        while self.accept("@"):
            name, args = self.ident(), []
            if self.accept("("):
                while not self.at(")"):
                    args.append(self.expr())
                    if not self.accept(","):
                        break
                self.expect(")")
            attrs.append((name, args))
        return attrs

    # -- types
    def type(self):
        name = self.ident()
        args = []
        if self.at("<") and name in TEMPLATED | {"var"}:
            self.i += 1
            while True:
                if self.peek().kind == "num":
                    args.append(self.primary())
                else:
                    args.append(self.type())
                if not self.accept(","):
                    break
            self.close_template()
        return Node("type", name=name, args=args)

    # -- module scope
    def module(self):
        items = []
        while self.peek().kind != "eof":
            if self.accept(";"):
                continue
            line = self.peek().line
            attrs = self.attributes()
            flag = None
            if self.at("virtual") or self.at("override"):
                flag = self.ident()
            if self.at("fn"):
                it = self.function()
                it.flag = flag
            elif self.at("struct"):
                it = self.struct()
            elif self.at("const"):
                self.i += 1
                name = self.ident()
                ty = self.type() if self.accept(":") else None
                self.expect("=")
                it = Node("const", name=name, type=ty, init=self.expr())
                self.expect(";")
            elif self.at("var"):
                self.i += 1
                space = []
                if self.accept("<"):
                    while not self.at(">"):
                        space.append(self.ident())
                        self.accept(",")
                    self.close_template()
                name = self.ident()
                self.expect(":")
                ty = self.type()
                init = self.expr() if self.accept("=") else None
                self.expect(";")
                it = Node("global", name=name, type=ty, space=space, init=init)
            else:
                t = self.peek()
                raise SyntaxError("%s:%d: unexpected %r at module scope" % (self.filename, t.line, t.text))
            it.attrs, it.line = attrs, line
            items.append(it)
        return items

    def struct(self):
        self.expect("struct")
        name = self.ident()
        self.expect("{")
        fields = []
        while not self.at("}"):
            self.attributes()
            fname = self.ident()
            self.expect(":")
            fields.append((fname, self.type()))
            if not self.accept(","):
                break
        self.expect("}")
        self.accept(";")
        return Node("struct", name=name, fields=fields)

    def function(self):
        self.expect("fn")
        name = self.ident()
        self.expect("(")
        params = []
        while not self.at(")"):
            pattrs = self.attributes()
            pname = self.ident()
            self.expect(":")
            params.append((pname, self.type(), pattrs))
            if not self.accept(","):
                break
        self.expect(")")
        ret = None
        if self.accept("->"):
            self.attributes()
            ret = self.type()
        return Node("fn", name=name, params=params, ret=ret, body=self.block(), flag=None)

    # -- statements
    def block(self):
        self.expect("{")
        stmts = []
        while not self.at("}"):
            stmts.append(self.statement())
        self.expect("}")
        return Node("block", stmts=stmts)

    def var_decl(self):
        kw = self.ident()  # let | var | const
        if kw == "var" and self.accept("<"):
            while not self.at(">"):
                self.ident()
                self.accept(",")
            self.close_template()
        name = self.ident()
        ty = self.type() if self.accept(":") else None
        init = self.expr() if self.accept("=") else None
        return Node("decl", kw=kw, name=name, type=ty, init=init)

    def simple_statement(self):
        """declaration, assignment, increment or call — without the trailing ';' (also used by for(;;))"""
        if self.at("let") or self.at("var") or self.at("const"):
            return self.var_decl()
        if self.at("_"):
            self.i += 1
            self.expect("=")
            return Node("exprstmt", expr=self.expr())
        lhs = self.unary()
        t = self.peek()
        if t.text in ("=", "+=", "-=", "*=", "/=", "%=", "&=", "|=", "^=", "<<=", ">>="):
            self.i += 1
            return Node("assign", op=t.text, lhs=lhs, rhs=self.expr())
        if t.text in ("++", "--"):
            self.i += 1
            return Node("assign", op=t.text[0] + "=", lhs=lhs, rhs=Node("num", text="1"))
        if lhs.kind != "call":
            raise SyntaxError("%s:%d: expression statement must be a call" % (self.filename, t.line))
        return Node("exprstmt", expr=lhs)

    def statement(self):
        line = self.peek().line
        st = self._statement()
        st.line = line
        return st

    def _statement(self):
        if self.accept(";"):
            return Node("block", stmts=[])
        if self.at("{"):
            return self.block()
        if self.accept("return"):
            e = None if self.at(";") else self.expr()
            self.expect(";")
            return Node("return", expr=e)
        if self.accept("if"):
            cond = self.expr()
            then = self.block()
            other = None
            if self.accept("else"):
                other = self.statement() if self.at("if") else self.block()
            return Node("if", cond=cond, then=then, other=other)
        if self.accept("for"):
            self.expect("(")
            init = None if self.at(";") else self.simple_statement()
            self.expect(";")
            cond = None if self.at(";") else self.expr()
            self.expect(";")
            update = None if self.at(")") else self.simple_statement()
            self.expect(")")
            return Node("for", init=init, cond=cond, update=update, body=self.block())
        if self.accept("while"):
            cond = self.expr()
            return Node("while", cond=cond, body=self.block())
        if self.accept("loop"):
            return Node("while", cond=Node("bool", value=True), body=self.block())
        if self.accept("switch"):
            sel = self.expr()
            self.expect("{")
            cases = []
            while not self.at("}"):
                selectors = []
                if self.accept("default"):
                    selectors.append(None)
                else:
                    self.expect("case")
                    while True:
                        selectors.append(None if self.accept("default") else self.expr())
                        if not self.accept(","):
                            break
                self.accept(":")
                cases.append((selectors, self.block()))
            self.expect("}")
            return Node("switch", sel=sel, cases=cases)
        if self.at("break") or self.at("continue"):
            kw = self.ident()
            self.expect(";")
            return Node("jump", kw=kw)
        st = self.simple_statement()
        self.expect(";")
        return st

    # -- expressions (precedence climbing; the C / WGSL binary precedence order)
    LEVELS = [["||"], ["&&"], ["|"], ["^"], ["&"], ["==", "!="], ["<", ">", "<=", ">="], ["<<", ">>"], ["+", "-"], ["*", "/", "%"]]

    def expr(self, level=0):
        if level == len(self.LEVELS):
            return self.unary()
        lhs = self.expr(level + 1)
        while self.peek().kind == "op" and self.peek().text in self.LEVELS[level]:
            op = self.peek().text
            self.i += 1
            lhs = Node("binary", op=op, lhs=lhs, rhs=self.expr(level + 1))
        return lhs

    def unary(self):
        t = self.peek()
        if t.kind == "op" and t.text in ("-", "!", "~", "*", "&"):
            self.i += 1
            return Node("unary", op=t.text, operand=self.unary())
        return self.postfix()

    def postfix(self):
        e = self.primary()
        while True:
            if self.accept("."):
                e = Node("member", base=e, name=self.ident())
            elif self.accept("["):
                e = Node("index", base=e, index=self.expr())
                self.expect("]")
            else:
                return e

    def args(self):
        self.expect("(")
        a = []
        while not self.at(")"):
            a.append(self.expr())
            if not self.accept(","):
                break
        self.expect(")")
        return a

    def primary(self):
        t = self.peek()
        if t.kind == "num":
            self.i += 1
            return Node("num", text=t.text)
        if self.accept("("):
            e = self.expr()
            self.expect(")")
            return Node("paren", expr=e)
        if self.at("true") or self.at("false"):
            return Node("bool", value=self.ident() == "true")
        if t.kind == "id":
            if t.text in TEMPLATED and self.peek(1).text == "<":
                ty = self.type()
                return Node("call", callee=None, type=ty, args=self.args())
            name = self.ident()
            if self.at("("):
                return Node("call", callee=name, type=None, args=self.args())
            return Node("ident", name=name)
        raise SyntaxError("%s:%d: unexpected %r in expression" % (self.filename, t.line, t.text))


# --------------------------------------------------------------------------------------------------------------
# emitter
# --------------------------------------------------------------------------------------------------------------

SCALARS = {"f32", "u32", "i32", "bool"}
# `const X = <abstract expression>;` without a type.  The WGSL specification keeps X abstract (so `1.0 + X` is evaluated in
# f64 and rounded once); naga 0.20 — the shader compiler behind bevy 0.14.0, which the reference pins (Cargo.toml:18) —
# evaluates the initialiser in abstract arithmetic and then CONCRETISES the constant at its declaration (abstract-typed
# const declarations arrived in later naga releases).  True = the pinned compiler.  The difference is observable in
# functions.wgsl:12,78 (`const C_SQR = 0.87 * 0.87; ... 1.0 + C_SQR ...`): 1 ULP in the denominator of the cube-sphere warp.
# WGSL2CPP_ABSTRACT_CONSTS=1 in the environment selects the specification's rule (tests report both).
CONST_DECLARATIONS_CONCRETIZE = os.environ.get("WGSL2CPP_ABSTRACT_CONSTS", "0") != "1"
BUILTIN_FUNCTIONS = {
    "abs", "all", "any", "ceil", "clamp", "cos", "cross", "distance", "dot", "exp", "exp2", "floor", "fract", "length", "log",
    "log2", "max", "min", "mix", "normalize", "pow", "round", "saturate", "select", "sign", "sin", "sqrt", "step", "tan",
    "transpose", "trunc", "pack2x16unorm", "pack4x8unorm", "unpack2x16unorm", "unpack4x8unorm", "textureLoad",
    "textureSampleLevel", "textureGather", "textureDimensions", "atomicAdd", "atomicSub", "atomicLoad", "atomicStore",
    "atomicExchange", "atomicMax", "atomicMin", "arrayLength", "countOneBits", "firstLeadingBit", "inverseSqrt", "fma",
    "smoothstep", "degrees", "radians", "storageBarrier", "workgroupBarrier",
}
SWIZZLE_RE = re.compile(r"^(?:[xyzw]{1,4}|[rgba]{1,4})$")


class Pipeline:
    def __init__(self, modules, entry_modules):
        self.modules = modules  # path -> Module (parsed: .items filled)
        self.entries = entry_modules
        self.overrides = {}  # (module path, name) of a virtual fn -> (module path, name) of its override
        self.field_names = set()
        for m in modules.values():
            for it in m.items.values():
                if it.kind == "struct":
                    self.field_names.update(f for f, _ in it.fields)
        # naga_oil composes the entry module with the modules it (transitively) imports; only those can override
        closure, todo = set(), [m.path for m in entry_modules]
        while todo:
            p = todo.pop()
            if p in closure or p not in modules:
                continue
            closure.add(p)
            todo += [src for src, _ in modules[p].imports.values()]
        for m in (modules[p] for p in sorted(closure)):
            for it in m.items.values():
                if it.kind == "fn" and it.flag == "override":
                    if it.name not in m.imports:
                        raise SyntaxError("override fn %s: nothing imported to override in %s" % (it.name, m.path))
                    self.overrides[m.imports[it.name]] = (m.path, it.name)
        self.emitted = {}  # (module path, name) -> C++ text
        self.order = []
        self.queue = []
        self.entry_points = []

    def resolve(self, module, name, want=True):
        """module-scope name as seen from `module` -> (defining module, item) or None"""
        seen = 0
        key = None
        if name in module.items:
            key = (module.path, name)
        elif name in module.imports:
            key = module.imports[name]
        if key is None:
            return None
        while key in self.overrides and seen < 8:
            key, seen = self.overrides[key], seen + 1
        mod = self.modules.get(key[0])
        if mod is None or key[1] not in mod.items:
            if not want:
                return None
            raise SyntaxError("%s: '%s' is imported from %s, which is not among the modules given (bevy-supplied? add it to "
                              "bevy_supplied.wgsl)" % (module.path, name, key[0]))
        if key not in self.emitted and key not in self.queue:
            self.queue.append(key)
        return mod, mod.items[key[1]]

    @staticmethod
    def mangle(mod, item):
        return "%s__%s" % (item.name, mod.tag)

    def run(self):
        for m in self.entries:
            for it in m.items.values():
                if it.kind == "fn" and any(a[0] == "compute" for a in it.attrs):
                    self.entry_points.append((m, it))
                    self.queue.append((m.path, it.name))
        while self.queue:
            key = self.queue.pop(0)
            if key in self.emitted:
                continue
            mod = self.modules[key[0]]
            item = mod.items[key[1]]
            self.emitted[key] = None  # cycle guard
            text = ItemEmitter(self, mod).item(item)
            self.emitted[key] = (item.kind, text, mod, item)
            self.order.append(key)


class ItemEmitter:
    def __init__(self, pipe, module):
        self.p, self.m = pipe, module
        self.scopes = []
        self.used = {}
        self.deps = []  # module-scope items this one's DECLARATION needs (for ordering structs / consts)

    # -- scopes
    def declare(self, name):
        # WGSL lets an inner declaration shadow an outer one AND read it in its own initialiser; C++ does not, so every
        # re-declaration of a name inside one function gets a C++ name of its own
        n = self.used.get(name, 0) + 1
        self.used[name] = n
        cpp = "l_" + name if n == 1 else "l_%s_%d" % (name, n)
        self.scopes[-1][name] = cpp
        return cpp

    def is_local(self, name):
        return any(name in s for s in self.scopes)

    def local(self, name):
        for s in reversed(self.scopes):
            if name in s:
                return s[name]
        raise KeyError(name)

    # -- types
    def const_int(self, e):
        if e.kind == "num":
            return int(re.sub(r"[iu]$", "", e.text), 0)
        if e.kind == "ident":
            r = self.p.resolve(self.m, e.name)
            if r and r[1].kind == "const":
                return ItemEmitter(self.p, r[0]).const_int(r[1].init)
        if e.kind == "type" and not e.args:
            return self.const_int(Node("ident", name=e.name))
        raise SyntaxError("array size must be a constant integer: %r" % e)

    def type(self, t):
        n, a = t.name, t.args
        if n in SCALARS:
            return n
        if n in ("vec2", "vec3", "vec4"):
            return "%s<%s>" % (n, self.type(a[0]))
        if re.match(r"mat[234]x[234]$", n):
            return "%s<%s>" % (n, self.type(a[0]))
        if n == "array":
            if len(a) == 2:
                return "warray<%s, %d>" % (self.type(a[0]), self.const_int(a[1]))
            return "rt_array<%s>" % self.type(a[0])
        if n == "atomic":
            return "w_atomic<%s>" % self.type(a[0])
        if n == "ptr":
            return "%s*" % self.type(a[-1] if a[-1].name not in ("read", "write", "read_write") else a[-2])
        if n in ("texture_2d", "texture_2d_array"):
            return "%s<%s>" % (n, self.type(a[0]))
        if n == "sampler":
            return "sampler"
        r = self.p.resolve(self.m, n)
        if r and r[1].kind == "struct":
            self.deps.append((r[0].path, r[1].name))
            return Pipeline.mangle(*r)
        raise SyntaxError("%s: unknown type %s" % (self.m.path, n))

    # -- expressions
    def num(self, text):
        if re.match(r"0[xX]", text):
            if text.endswith("u"):
                return "u32(%su)" % text[:-1]
            if text.endswith("i"):
                return "i32(%s)" % text[:-1]
            return "AI(%s)" % text
        suffix = text[-1] if text[-1] in "fhiu" else ""
        body = text[: len(text) - len(suffix)]
        is_float = "." in body or "e" in body.lower()
        if suffix == "u":
            return "u32(%su)" % body
        if suffix == "i":
            return "i32(%s)" % body
        if suffix in ("f", "h"):
            return "f32(%sf)" % (body if is_float else body + ".0")
        if is_float:
            return "AF(%s)" % body
        return "AI(%sLL)" % body

    def expr(self, e):
        k = e.kind
        if k == "num":
            return self.num(e.text)
        if k == "bool":
            return "true" if e.value else "false"
        if k == "paren":
            return "(%s)" % self.expr(e.expr)
        if k == "ident":
            if self.is_local(e.name):
                return self.local(e.name)
            r = self.p.resolve(self.m, e.name)
            if r is None:
                raise SyntaxError("%s: unknown identifier %s" % (self.m.path, e.name))
            if r[1].kind == "const":
                self.deps.append((r[0].path, r[1].name))
            return Pipeline.mangle(*r)
        if k == "unary":
            if e.op == "&":
                return "(&(%s))" % self.expr(e.operand)
            if e.op == "*":
                return "(*(%s))" % self.expr(e.operand)
            return "(%s(%s))" % (e.op, self.expr(e.operand))
        if k == "binary":
            a, b = self.expr(e.lhs), self.expr(e.rhs)
            fn = {"/": "w_div", "%": "w_mod", "<<": "w_shl", ">>": "w_shr"}.get(e.op)
            if fn:
                return "%s(%s, %s)" % (fn, a, b)
            return "(%s %s %s)" % (a, e.op, b)
        if k == "index":
            return "%s[%s]" % (self.expr(e.base), self.expr(e.index))
        if k == "member":
            base = self.expr(e.base)
            if SWIZZLE_RE.match(e.name):
                comps = ["xyzwrgba".index(c) % 4 for c in e.name]
                if len(comps) == 1:
                    return "%s.%s" % (base, e.name if e.name in self.p.field_names else "xyzw"[comps[0]])
                if e.name in self.p.field_names:
                    # `.xy` is both a swizzle and a struct field name in these shaders: the C++ side picks by type
                    return "w_member_or_swizzle_%s(%s)" % (e.name, base)
                return "w_swizzle<%s>(%s)" % (", ".join(map(str, comps)), base)
            return "%s.%s" % (base, e.name)
        if k == "call":
            args = ", ".join(self.expr(a) for a in e.args)
            if e.type is not None:
                n = e.type.name
                if n == "array":
                    return "%s{{%s}}" % (self.type(e.type), args) if e.args else "%s{}" % self.type(e.type)
                if n == "bitcast":
                    return "w_bitcast<%s>(%s)" % (self.type(e.type.args[0]), args)
                return "%s(%s)" % (self.type(e.type), args)
            n = e.callee
            if not self.is_local(n):
                r = self.p.resolve(self.m, n, want=False)
                if r is not None:
                    if r[1].kind == "struct":
                        self.deps.append((r[0].path, r[1].name))
                        return "%s{%s}" % (Pipeline.mangle(*r), args)
                    if r[1].kind == "fn":
                        return "%s(%s)" % (Pipeline.mangle(*r), args)
                    raise SyntaxError("%s: %s is not callable" % (self.m.path, n))
            if n in SCALARS:
                return "w_cast<%s>(%s)" % (n, args) if e.args else "%s{}" % n
            if n in ("vec2", "vec3", "vec4"):
                return "mkvec<%s>(%s)" % (n[3], args)
            if n == "array":
                return "mkarray(%s)" % args
            if n in BUILTIN_FUNCTIONS:
                return "w_%s(%s)" % (n, args)
            # an import that cannot be resolved (module absent) gives the precise error
            self.p.resolve(self.m, n)
            raise SyntaxError("%s: unknown function %s" % (self.m.path, n))
        raise SyntaxError("cannot emit expression %r" % e)

    # -- statements
    def stmt(self, s, ind):
        pad = "    " * ind
        k = s.kind
        if k == "block":
            self.scopes.append({})
            body = "".join(self.stmt(x, ind + 1) for x in s.stmts)
            self.scopes.pop()
            return "%s{\n%s%s}\n" % (pad, body, pad)
        if k == "decl":
            return pad + self.decl(s) + ";\n"
        if k == "assign":
            return pad + self.assign(s) + ";\n"
        if k == "exprstmt":
            return "%s%s;\n" % (pad, self.expr(s.expr))
        if k == "return":
            return "%sreturn%s;\n" % (pad, " " + self.expr(s.expr) if s.expr is not None else "")
        if k == "jump":
            return "%s%s;\n" % (pad, s.kw)
        if k == "if":
            out = "%sif (w_cond(%s))\n%s" % (pad, self.expr(s.cond), self.stmt(s.then, ind))
            if s.other is not None:
                out += "%selse\n%s" % (pad, self.stmt(s.other, ind))
            return out
        if k == "while":
            return "%swhile (w_cond(%s))\n%s" % (pad, self.expr(s.cond), self.stmt(s.body, ind))
        if k == "for":
            self.scopes.append({})
            init = self.decl(s.init) if s.init is not None and s.init.kind == "decl" else (self.simple(s.init) if s.init is not None else "")
            cond = "w_cond(%s)" % self.expr(s.cond) if s.cond is not None else ""
            update = self.simple(s.update) if s.update is not None else ""
            out = "%sfor (%s; %s; %s)\n%s" % (pad, init, cond, update, self.stmt(s.body, ind))
            self.scopes.pop()
            return out
        if k == "switch":
            out = "%sswitch (w_switch(%s)) {\n" % (pad, self.expr(s.sel))
            for selectors, body in s.cases:
                for sel in selectors:
                    out += "%s%s:\n" % (pad, "default" if sel is None else "case %d" % self.const_int(sel))
                out += self.stmt(body, ind + 1) + "%s    break;\n" % pad
            return out + "%s}\n" % pad
        raise SyntaxError("cannot emit statement %r" % s)

    def simple(self, s):
        if s.kind == "decl":
            return self.decl(s)
        if s.kind == "assign":
            return self.assign(s)
        return self.expr(s.expr)

    def decl(self, s):
        init = self.expr(s.init) if s.init is not None else None  # the initialiser sees the OUTER meaning of the name
        name = self.declare(s.name)
        if s.kw == "const":
            return "const auto %s = %s" % (name, init) if s.type is None else "const %s %s = %s" % (self.type(s.type), name, init)
        const = "const " if s.kw == "let" else ""
        if s.type is not None:
            return "%s%s %s%s" % (const, self.type(s.type), name, " = %s" % init if init is not None else "{}")
        return "%sauto %s = w_concretize(%s)" % (const, name, init)

    def assign(self, s):
        lhs, rhs = self.expr(s.lhs), self.expr(s.rhs)
        if s.op == "=":
            return "%s = %s" % (lhs, rhs)
        op = s.op[:-1]
        fn = {"/": "w_div", "%": "w_mod", "<<": "w_shl", ">>": "w_shr"}.get(op)
        if fn:
            return "w_assign(%s, %s(%s, %s))" % (lhs, fn, lhs, rhs)
        return "w_assign(%s, (%s %s %s))" % (lhs, lhs, op, rhs)

    # -- items
    def item(self, it):
        name = Pipeline.mangle(self.m, it)
        where = "// %s:%d" % (self.m.filename, it.line)
        if it.kind == "struct":
            fields = "".join("    %s %s{};\n" % (self.type(t), f) for f, t in it.fields)
            return "%s\nstruct %s {\n%s};\n" % (where, name, fields), list(self.deps)
        if it.kind == "const":
            init = self.expr(it.init)
            ty = self.type(it.type) if it.type is not None else "auto"
            if it.type is None and CONST_DECLARATIONS_CONCRETIZE:
                init = "w_concretize(%s)" % init
            return "%s\nstatic inline const %s %s = %s;\n" % (where, ty, name, init), list(self.deps)
        if it.kind == "global":
            space = ",".join(it.space) or "handle"
            return "%s  var<%s>\n%s %s{};\n" % (where, space, self.type(it.type), name), list(self.deps)
        if it.kind == "fn":
            self.scopes.append({})
            params = ", ".join("%s %s" % (self.type(t), self.declare(n)) for n, t, _ in it.params)
            ret = self.type(it.ret) if it.ret is not None else "void"
            body = self.stmt(it.body, 0)
            self.scopes.pop()
            wg = [a for a in it.attrs if a[0] == "workgroup_size"]
            extra = ""
            if wg:
                sizes = [self.const_int(x) for x in wg[0][1]] + [1, 1]
                extra = "static constexpr unsigned %s_workgroup_size[3] = {%d, %d, %d};\n" % (name, sizes[0], sizes[1], sizes[2])
            return "%s\n%s%s %s(%s)\n%s" % (where, extra, ret, name, params, body), list(self.deps)
        raise SyntaxError("cannot emit item %r" % it)


# --------------------------------------------------------------------------------------------------------------
# driver
# --------------------------------------------------------------------------------------------------------------


def load_modules(files, defs):
    modules = {}
    for fn in files:
        with open(fn) as f:
            raw = f.read()
        stem = os.path.splitext(os.path.basename(fn))[0]
        mod = Module(stem, fn, preprocess(raw, defs))
        for it in Parser(lex(mod.text), fn).module():
            mod.items[it.name] = it
        modules[mod.path] = mod
    return modules


def translate(struct_name, defs, entry_files, search):
    files = list(entry_files)
    for d in search:
        if os.path.isdir(d):
            files += sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(".wgsl") and os.path.join(d, f) not in files)
        elif d not in files:
            files.append(d)
    modules = load_modules(files, defs)
    by_file = {m.filename: m for m in modules.values()}
    pipe = Pipeline(modules, [by_file[f] for f in entry_files])
    pipe.run()

    # order: structs and consts depth-first by what their declarations need; then globals; then functions
    out, done = [], set()

    def emit_decl(key):
        if key in done:
            return
        done.add(key)
        kind, (text, deps), _, _ = pipe.emitted[key]
        for d in deps:
            if pipe.emitted[d][0] in ("struct", "const"):
                emit_decl(d)
        out.append(text)

    for key in pipe.order:
        if pipe.emitted[key][0] in ("struct", "const"):
            emit_decl(key)
    for key in pipe.order:
        if pipe.emitted[key][0] == "global":
            out.append(pipe.emitted[key][1][0])
    for key in pipe.order:
        if pipe.emitted[key][0] == "fn":
            out.append(pipe.emitted[key][1][0])
    head = ("// GENERATED by oracle/wgsl_ref/wgsl2cpp.py — do not edit, do not commit (oracle/_ref/ is git-ignored).\n"
            "// A mechanical translation of the reference's WGSL, read where it lies:\n"
            + "".join("//   %s\n" % f for f in sorted({pipe.emitted[k][2].filename for k in pipe.order}))
            + "// shader defs: %s\n" % (", ".join(sorted(defs)) or "(none)"))
    body = "\n".join(out)
    body = "\n".join("    " + line if line else "" for line in body.split("\n"))
    return "%sstruct %s {\n%s\n};\n" % (head, struct_name, body)


def main(argv):
    if len(argv) < 5:
        sys.exit(__doc__)
    out_path, struct_name, defs, entries = argv[1], argv[2], argv[3], argv[4].split(",")
    defs = set() if defs == "-" else set(defs.split(","))
    text = translate(struct_name, defs, entries, argv[5:])
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    with open(out_path, "w") as f:
        f.write(text)


if __name__ == "__main__":
    main(sys.argv)
