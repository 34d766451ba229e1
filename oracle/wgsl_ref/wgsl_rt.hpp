// wgsl_rt.hpp — the WGSL type system and built-in functions as C++ (TEST INFRASTRUCTURE, oracle/ only).
//
// The code wgsl2cpp.py emits is the reference's shader text with C++ punctuation; this header makes that text mean
// what WGSL says it means:
//   * f32 / u32 / i32 / bool are the machine types; every f32 operation is ONE IEEE binary32 operation (the build
//     uses -ffp-contract=off, no fast-math), integer arithmetic wraps (-fwrapv), shifts take the amount mod 32,
//     integer division by zero returns the dividend (WGSL §expressions) instead of trapping;
//   * AI / AF are WGSL's abstract-int / abstract-float: literals without a suffix and constant expressions of them
//     are evaluated in i64 / f64 and become concrete (i32 / u32 / f32) only when they meet a concrete operand, are
//     assigned, or are bound by `let` / `var` (w_concretize) — this is what naga's constant evaluator does, and it is
//     observable: `1.0 + C_SQR` with `const C_SQR = 0.87 * 0.87` is rounded ONCE from f64;
//   * vecN<T>, matCxR<T> with component-wise operators and scalar broadcast, swizzles, constructors that flatten
//     their arguments; array<T, N> (value type) and runtime-sized storage arrays;
//   * built-in functions as the WGSL specification defines them.  Where the specification leaves precision to the
//     implementation (sqrt, pow, log2, division, normalize, length, distance, mix, matrix * vector) this file uses
//     the correctly rounded / textbook form: sqrtf, powf (exact for 2^integer), one IEEE division, length =
//     sqrt(x*x + y*y + z*z) summed left to right, normalize = v / length(v), mix = a*(1-t) + b*t, M*v = sum of
//     column_i * v_i left to right.  These are harness choices (what a GPU driver supplies), stated here once.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <type_traits>

namespace wgsl {

using f32 = float;
using u32 = uint32_t;
using i32 = int32_t;

// ---------------------------------------------------------------------------------------------------- scalars
struct AI;
struct AF;
template <class T> inline constexpr bool is_abs = std::is_same_v<T, AI> || std::is_same_v<T, AF>;
template <class T> inline constexpr bool is_conc = std::is_same_v<T, f32> || std::is_same_v<T, u32> || std::is_same_v<T, i32> || std::is_same_v<T, bool>;
template <class T> inline constexpr bool is_scalar = is_abs<T> || is_conc<T>;

template <class T, class S> constexpr T w_cast(S s);

struct AI {
    long long v = 0;
    constexpr AI() = default;
    constexpr explicit AI(long long x) : v(x) {}
    template <class T, std::enable_if_t<is_conc<T>, int> = 0> constexpr operator T() const { return static_cast<T>(v); }
};
struct AF {
    double v = 0;
    constexpr AF() = default;
    constexpr explicit AF(double x) : v(x) {}
    constexpr AF(AI i) : v(double(i.v)) {}
    template <class T, std::enable_if_t<std::is_same_v<T, f32>, int> = 0> constexpr operator T() const { return static_cast<f32>(v); }
};

// the common type of two scalar operands (WGSL automatic conversions: abstract -> concrete, AI -> AF)
template <class A, class B> struct common { using type = void; };
template <class A> struct common<A, A> { using type = A; };
template <> struct common<AI, AF> { using type = AF; };
template <> struct common<AF, AI> { using type = AF; };
template <> struct common<AI, f32> { using type = f32; };  template <> struct common<f32, AI> { using type = f32; };
template <> struct common<AI, u32> { using type = u32; };  template <> struct common<u32, AI> { using type = u32; };
template <> struct common<AI, i32> { using type = i32; };  template <> struct common<i32, AI> { using type = i32; };
template <> struct common<AF, f32> { using type = f32; };  template <> struct common<f32, AF> { using type = f32; };
template <class A, class B> using common_t = typename common<A, B>::type;
template <class A, class... R> struct common_n { using type = A; };
template <class A, class B, class... R> struct common_n<A, B, R...> { using type = typename common_n<common_t<A, B>, R...>::type; };

// value conversion T(e) — WGSL §conversion expressions: f32 -> integer truncates and saturates, integer <-> integer
// keeps the bits, anything -> bool is `!= 0`
template <class T, class S> constexpr T w_cast(S s) {
    if constexpr (std::is_same_v<T, S>) return s;
    else if constexpr (std::is_same_v<T, AI>) { static_assert(std::is_same_v<S, AI>); return s; }
    else if constexpr (std::is_same_v<T, AF>) { static_assert(is_abs<S>); return AF(s); }
    else if constexpr (is_abs<S>) {
        if constexpr (std::is_same_v<T, bool>) return s.v != 0;
        else if constexpr (std::is_same_v<S, AF> && std::is_integral_v<T>) return w_cast<T>(static_cast<f32>(s.v));
        else return static_cast<T>(s.v);
    }
    else if constexpr (std::is_same_v<T, bool>) return s != S(0);
    else if constexpr (std::is_same_v<S, f32> && std::is_same_v<T, u32>) { return !(s > 0.0f) ? 0u : (s >= 4294967296.0f ? 0xFFFFFFFFu : static_cast<u32>(s)); }
    else if constexpr (std::is_same_v<S, f32> && std::is_same_v<T, i32>) { return s != s ? 0 : (s <= -2147483648.0f ? INT32_MIN : (s >= 2147483648.0f ? INT32_MAX : static_cast<i32>(s))); }
    else return static_cast<T>(s);
}

// ---------------------------------------------------------------------------------------------------- vectors
template <class T, int N> struct vec;
template <class T> inline constexpr bool is_vec = false;
template <class T, int N> inline constexpr bool is_vec<vec<T, N>> = true;
template <class T> struct elem_of { using type = T; static constexpr int dim = 0; };
template <class T, int N> struct elem_of<vec<T, N>> { using type = T; static constexpr int dim = N; };
template <class T> using elem_t = typename elem_of<T>::type;
template <class T> inline constexpr int dim_v = elem_of<T>::dim;
template <class T> inline constexpr int ncomp = dim_v<T> == 0 ? 1 : dim_v<T>;
template <class T> inline constexpr bool is_value = is_scalar<T> || is_vec<T>;

template <class T, int N> struct vec_storage;
template <class T> struct vec_storage<T, 2> { T x{}, y{}; };
template <class T> struct vec_storage<T, 3> { T x{}, y{}, z{}; };
template <class T> struct vec_storage<T, 4> { T x{}, y{}, z{}, w{}; };

template <class T, int N> struct vec : vec_storage<T, N> {
    using elem = T;
    static constexpr int dim = N;
    constexpr vec() = default;
    // T(e): splat; T(e1, .., eN) and mixed vector / scalar forms: the arguments' components, flattened, converted
    template <class... A, std::enable_if_t<(sizeof...(A) >= 1) && (is_value<A> && ...), int> = 0>
    constexpr explicit(!(sizeof...(A) == 1 && ((is_vec<A> && is_abs<elem_t<A>>) && ...))) vec(const A&... a) {
        constexpr int total = (ncomp<A> + ...);
        static_assert(total == N || (total == 1 && sizeof...(A) == 1), "vector constructor: component count");
        if constexpr (total == 1) { for (int i = 0; i < N; i++) (*this)[i] = (w_cast<T>(a), ...); }
        else { int k = 0; (put(k, a), ...); }
    }
    constexpr T& operator[](int i) { return (&this->x)[i]; }
    constexpr const T& operator[](int i) const { return (&this->x)[i]; }
    template <class I, std::enable_if_t<is_scalar<I> && !std::is_same_v<I, int>, int> = 0> constexpr T& operator[](I i) { return (&this->x)[clampi(i)]; }
    template <class I, std::enable_if_t<is_scalar<I> && !std::is_same_v<I, int>, int> = 0> constexpr const T& operator[](I i) const { return (&this->x)[clampi(i)]; }

private:
    template <class I> static constexpr int clampi(I i) { long long v = w_cast<i32>(i); return int(v < 0 ? 0 : (v >= N ? N - 1 : v)); }
    template <class A> constexpr void put(int& k, const A& a) {
        if constexpr (is_vec<A>) { for (int i = 0; i < A::dim; i++) (*this)[k++] = w_cast<T>(a[i]); }
        else (*this)[k++] = w_cast<T>(a);
    }
};
template <class T> using vec2 = vec<T, 2>;
template <class T> using vec3 = vec<T, 3>;
template <class T> using vec4 = vec<T, 4>;

// vecN(e1, ..) without a type: the component type is inferred
template <int N, class... A> constexpr auto mkvec(const A&... a) {
    using T = typename common_n<elem_t<A>...>::type;
    static_assert(!std::is_void_v<T>, "vecN(...): components have no common type");
    return vec<T, N>(a...);
}
template <int... C, class V> constexpr auto w_swizzle(const V& v) { return vec<elem_t<V>, int(sizeof...(C))>(v[C]...); }
// `.xy` where some struct of the shader has a FIELD called xy: field if there is one, swizzle otherwise
template <class E> constexpr decltype(auto) w_member_or_swizzle_xy(E&& e) {
    if constexpr (requires { e.xy; }) return (e.xy); else return w_swizzle<0, 1>(e);
}

// ---------------------------------------------------------------------------------------------------- operators
template <class T, class S> constexpr auto conv(const S& s) { return w_cast<T>(s); }

// apply f component-wise over scalars / vectors of a common element type
template <class F, class... A> constexpr auto w_map(F f, const A&... a) {
    using C = typename common_n<elem_t<A>...>::type;
    static_assert(!std::is_void_v<C>, "operands have no common type");
    constexpr int N = std::max({dim_v<A>...});
    static_assert(((dim_v<A> == 0 || dim_v<A> == N) && ...), "vector sizes differ");
    auto get = [](const auto& x, int i) { if constexpr (is_vec<std::decay_t<decltype(x)>>) return conv<C>(x[i]); else return conv<C>(x); };
    if constexpr (N == 0) return f(conv<C>(a)...);
    else {
        using R = decltype(f(get(a, 0)...));
        vec<R, N> r;
        for (int i = 0; i < N; i++) r[i] = f(get(a, i)...);
        return r;
    }
}
template <class T> constexpr auto raw(T x) { if constexpr (is_abs<T>) return x.v; else return x; }
template <class T, class R> constexpr auto rewrap(R r) { if constexpr (is_abs<T> && !std::is_same_v<R, bool>) return T(r); else return r; }

// participates when an operand is abstract or a vector (plain concrete scalars use the machine operators)
template <class A, class B> inline constexpr bool op_applies = is_value<A> && is_value<B> && (is_abs<A> || is_abs<B> || is_vec<A> || is_vec<B>);
#define WGSL_BINARY(op)                                                                                              \
    template <class A, class B, std::enable_if_t<op_applies<A, B>, int> = 0> constexpr auto operator op(const A& a, const B& b) { \
        return w_map([](auto x, auto y) { using T = decltype(x); if constexpr (std::is_same_v<T, bool>) return bool(x op y); else return rewrap<T>(static_cast<std::conditional_t<is_abs<T>, decltype(raw(x) op raw(y)), std::conditional_t<std::is_same_v<decltype(raw(x) op raw(y)), bool>, bool, T>>>(raw(x) op raw(y))); }, a, b); \
    }
WGSL_BINARY(+) WGSL_BINARY(-) WGSL_BINARY(*) WGSL_BINARY(&) WGSL_BINARY(|) WGSL_BINARY(^)
WGSL_BINARY(==) WGSL_BINARY(!=) WGSL_BINARY(<) WGSL_BINARY(>) WGSL_BINARY(<=) WGSL_BINARY(>=)
#undef WGSL_BINARY
template <class A, std::enable_if_t<is_abs<A> || is_vec<A>, int> = 0> constexpr auto operator-(const A& a) { return w_map([](auto x) { return rewrap<decltype(x)>(static_cast<decltype(raw(x))>(-raw(x))); }, a); }
template <class A, std::enable_if_t<is_abs<A> || is_vec<A>, int> = 0> constexpr auto operator~(const A& a) { return w_map([](auto x) { return rewrap<decltype(x)>(static_cast<decltype(raw(x))>(~raw(x))); }, a); }
template <class A, std::enable_if_t<is_vec<A>, int> = 0> constexpr auto operator!(const A& a) { return w_map([](bool x) { return !x; }, a); }

// e1 / e2, e1 % e2 (WGSL: integer x/0 = x, x%0 = 0, i32 MIN/-1 = MIN; f32 % is x - y * trunc(x / y))
template <class T> constexpr T div1(T x, T y) {
    if constexpr (std::is_same_v<T, f32>) return x / y;
    else if constexpr (is_abs<T>) return T(x.v / y.v);
    else if constexpr (std::is_same_v<T, i32>) return (y == 0 || (x == INT32_MIN && y == -1)) ? x : x / y;
    else return y == 0 ? x : x / y;
}
template <class T> constexpr T mod1(T x, T y) {
    if constexpr (std::is_same_v<T, f32>) return x - y * std::trunc(x / y);
    else if constexpr (std::is_same_v<T, AF>) return AF(x.v - y.v * std::trunc(x.v / y.v));
    else if constexpr (std::is_same_v<T, AI>) return AI(x.v % y.v);
    else if constexpr (std::is_same_v<T, i32>) return (y == 0 || (x == INT32_MIN && y == -1)) ? 0 : x % y;
    else return y == 0 ? T(0) : x % y;
}
template <class A, class B> constexpr auto w_div(const A& a, const B& b) { return w_map([](auto x, auto y) { return div1(x, y); }, a, b); }
template <class A, class B> constexpr auto w_mod(const A& a, const B& b) { return w_map([](auto x, auto y) { return mod1(x, y); }, a, b); }
// shifts: the amount is u32 (or abstract), taken modulo the bit width; the result has the type of the left operand
template <class A, class B> constexpr auto w_shl(const A& a, const B& b) {
    if constexpr (is_vec<A>) { A r; for (int i = 0; i < A::dim; i++) r[i] = w_shl(a[i], [&] { if constexpr (is_vec<B>) return b[i]; else return b; }()); return r; }
    else if constexpr (std::is_same_v<A, AI>) { if constexpr (is_abs<B>) return AI(a.v << (w_cast<u32>(b) & 63u)); else return w_shl(w_cast<i32>(a), b); }
    else return A(u32(a) << (w_cast<u32>(b) & 31u));
}
template <class A, class B> constexpr auto w_shr(const A& a, const B& b) {
    if constexpr (is_vec<A>) { A r; for (int i = 0; i < A::dim; i++) r[i] = w_shr(a[i], [&] { if constexpr (is_vec<B>) return b[i]; else return b; }()); return r; }
    else if constexpr (std::is_same_v<A, AI>) { if constexpr (is_abs<B>) return AI(a.v >> (w_cast<u32>(b) & 63u)); else return w_shr(w_cast<i32>(a), b); }
    else return A(a >> (w_cast<u32>(b) & 31u));  // arithmetic for i32, logical for u32
}

// `let x = e;` / `var x = e;` without a type: abstract values become i32 / f32
template <class T> constexpr auto w_concretize(const T& v) {
    if constexpr (std::is_same_v<T, AI>) return w_cast<i32>(v);
    else if constexpr (std::is_same_v<T, AF>) return w_cast<f32>(v);
    else if constexpr (is_vec<T>) { if constexpr (std::is_same_v<elem_t<T>, AI>) return vec<i32, T::dim>(v); else if constexpr (std::is_same_v<elem_t<T>, AF>) return vec<f32, T::dim>(v); else return v; }
    else return v;
}
template <class L, class R> constexpr void w_assign(L& l, const R& r) { l = r; }
constexpr bool w_cond(bool b) { return b; }
template <class T> constexpr long long w_switch(T v) { if constexpr (is_abs<T>) return v.v; else return (long long)v; }

// ---------------------------------------------------------------------------------------------------- arrays
template <class T, int N> struct warray {
    std::array<T, N> a{};
    template <class I> constexpr T& operator[](I i) { return a[clampi(i)]; }
    template <class I> constexpr const T& operator[](I i) const { return a[clampi(i)]; }
    template <class I> static constexpr int clampi(I i) { long long v; if constexpr (is_abs<I>) v = i.v; else v = (long long)i; return int(v < 0 ? 0 : (v >= N ? N - 1 : v)); }
};
template <class... A> constexpr auto mkarray(const A&... a) {
    using T = std::common_type_t<decltype(w_concretize(a))...>;
    return warray<T, int(sizeof...(A))>{{{T(w_concretize(a))...}}};
}
// runtime-sized storage array: an out-of-range access sets `overflow` and lands in a scratch element (robust buffer
// access discards / clamps it on a GPU; the harness reports it)
template <class T> struct rt_array {
    T* data = nullptr;
    size_t size = 0;
    bool overflow = false;
    T scratch{};
    template <class I> T& operator[](I i) {
        long long v; if constexpr (is_abs<I>) v = i.v; else v = (long long)i;
        if (v < 0 || size_t(v) >= size) { overflow = true; scratch = T{}; return scratch; }
        return data[v];
    }
};
template <class T> struct w_atomic { T v{}; };
template <class T> T w_atomicLoad(w_atomic<T>* p) { return p->v; }
template <class T, class V> void w_atomicStore(w_atomic<T>* p, V v) { p->v = w_cast<T>(v); }
template <class T, class V> T w_atomicAdd(w_atomic<T>* p, V v) { T old = p->v; p->v = T(old + w_cast<T>(v)); return old; }
template <class T, class V> T w_atomicSub(w_atomic<T>* p, V v) { T old = p->v; p->v = T(old - w_cast<T>(v)); return old; }
template <class T, class V> T w_atomicExchange(w_atomic<T>* p, V v) { T old = p->v; p->v = w_cast<T>(v); return old; }

// ---------------------------------------------------------------------------------------------------- matrices
template <int C, int R, class T> struct mat {
    vec<T, R> col[C];
    constexpr mat() = default;
    template <class... A, std::enable_if_t<(sizeof...(A) >= 1), int> = 0> constexpr explicit mat(const A&... a) {
        constexpr int total = (ncomp<A> + ...);
        static_assert(total == C * R, "matrix constructor: component count");
        int k = 0;
        (put(k, a), ...);
    }
    template <class I> constexpr vec<T, R>& operator[](I i) { return col[w_cast<i32>(i)]; }
    template <class I> constexpr const vec<T, R>& operator[](I i) const { return col[w_cast<i32>(i)]; }

private:
    template <class A> constexpr void put(int& k, const A& a) {
        if constexpr (is_vec<A>) { for (int i = 0; i < A::dim; i++, k++) col[k / R][k % R] = w_cast<T>(a[i]); }
        else { col[k / R][k % R] = w_cast<T>(a); k++; }
    }
};
#define WGSL_MAT(C, R) template <class T> using mat##C##x##R = mat<C, R, T>;
WGSL_MAT(2, 2) WGSL_MAT(2, 3) WGSL_MAT(2, 4) WGSL_MAT(3, 2) WGSL_MAT(3, 3) WGSL_MAT(3, 4) WGSL_MAT(4, 2) WGSL_MAT(4, 3) WGSL_MAT(4, 4)
#undef WGSL_MAT
template <int C, int R, class T> constexpr mat<R, C, T> w_transpose(const mat<C, R, T>& m) {
    mat<R, C, T> t;
    for (int c = 0; c < C; c++) for (int r = 0; r < R; r++) t.col[r][c] = m.col[c][r];
    return t;
}
// M * v = col_0 * v_0 + col_1 * v_1 + ... (left to right)
template <int C, int R, class T> constexpr vec<T, R> operator*(const mat<C, R, T>& m, const vec<T, C>& v) {
    vec<T, R> r = m.col[0] * v[0];
    for (int c = 1; c < C; c++) r = r + m.col[c] * v[c];
    return r;
}

// ---------------------------------------------------------------------------------------------------- built-ins
template <class T> inline constexpr bool is_floaty = std::is_same_v<T, f32> || std::is_same_v<T, AF>;
#define WGSL_FN1(name, body) template <class A> constexpr auto w_##name(const A& a) { return w_map([](auto x) { using T = decltype(x); (void)sizeof(T); body; }, a); }
#define WGSL_FN2(name, body) template <class A, class B> constexpr auto w_##name(const A& a, const B& b) { return w_map([](auto x, auto y) { using T = decltype(x); (void)sizeof(T); body; }, a, b); }
#define WGSL_FN3(name, body) template <class A, class B, class C> constexpr auto w_##name(const A& a, const B& b, const C& c) { return w_map([](auto x, auto y, auto z) { using T = decltype(x); (void)sizeof(T); body; }, a, b, c); }
template <class T> constexpr T flt(T x) { if constexpr (std::is_same_v<T, AI>) return x; else return x; }
// float built-ins concretise an abstract argument to f32 first only when it must meet an f32; a purely abstract call
// (`sqrt(2.0)` in a const) stays f64
template <class T, class F> constexpr T fl1(T x, F f) { if constexpr (std::is_same_v<T, AF>) return AF(f(x.v)); else if constexpr (std::is_same_v<T, AI>) return T(f(double(x.v))); else return f(x); }
WGSL_FN1(abs, if constexpr (is_abs<T>) return T(x.v < 0 ? -x.v : x.v); else if constexpr (std::is_same_v<T, f32>) return std::fabs(x); else if constexpr (std::is_same_v<T, u32>) return x; else return T(x < 0 ? -x : x))
WGSL_FN1(sqrt, if constexpr (std::is_same_v<T, f32>) return std::sqrt(x); else return AF(std::sqrt(AF(x).v)))
WGSL_FN1(floor, if constexpr (std::is_same_v<T, f32>) return std::floor(x); else return AF(std::floor(AF(x).v)))
WGSL_FN1(ceil, if constexpr (std::is_same_v<T, f32>) return std::ceil(x); else return AF(std::ceil(AF(x).v)))
WGSL_FN1(trunc, if constexpr (std::is_same_v<T, f32>) return std::trunc(x); else return AF(std::trunc(AF(x).v)))
WGSL_FN1(fract, if constexpr (std::is_same_v<T, f32>) return x - std::floor(x); else return AF(AF(x).v - std::floor(AF(x).v)))
WGSL_FN1(log2, if constexpr (std::is_same_v<T, f32>) return std::log2(x); else return AF(std::log2(AF(x).v)))
WGSL_FN1(saturate, if constexpr (std::is_same_v<T, f32>) return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); else return AF(AF(x).v < 0 ? 0 : (AF(x).v > 1 ? 1 : AF(x).v)))
WGSL_FN2(min, if constexpr (is_abs<T>) return T(y.v < x.v ? y.v : x.v); else return y < x ? y : x)
WGSL_FN2(max, if constexpr (is_abs<T>) return T(x.v < y.v ? y.v : x.v); else return x < y ? y : x)
WGSL_FN2(step, if constexpr (is_abs<T>) return AF(x.v <= y.v ? 1.0 : 0.0); else return x <= y ? T(1) : T(0))  // step(edge, x): 1 if edge <= x
WGSL_FN2(pow, if constexpr (std::is_same_v<T, f32>) return std::pow(x, y); else return AF(std::pow(AF(x).v, AF(y).v)))
WGSL_FN3(clamp, if constexpr (is_abs<T>) return T(x.v < y.v ? y.v : (x.v > z.v ? z.v : x.v)); else { T lo = x < y ? y : x; return z < lo ? z : lo; })  // min(max(e, low), high)
WGSL_FN3(mix, if constexpr (std::is_same_v<T, f32>) return x * (1.0f - z) + y * z; else return AF(AF(x).v * (1.0 - AF(z).v) + AF(y).v * AF(z).v))
#undef WGSL_FN1
#undef WGSL_FN2
#undef WGSL_FN3
template <class F, class Tt, class C> constexpr auto w_select(const F& f, const Tt& t, const C& cond) {
    if constexpr (is_vec<C>) { auto r = w_map([](auto x, auto) { return x; }, f, t); auto tt = w_map([](auto, auto y) { return y; }, f, t); for (int i = 0; i < C::dim; i++) if (cond[i]) r[i] = tt[i]; return r; }
    else { auto ff = w_map([](auto x, auto) { return x; }, f, t); auto tt = w_map([](auto, auto y) { return y; }, f, t); return cond ? tt : ff; }
}
template <class V> constexpr bool w_all(const V& v) { if constexpr (is_vec<V>) { bool r = true; for (int i = 0; i < V::dim; i++) r = r && v[i]; return r; } else return v; }
template <class V> constexpr bool w_any(const V& v) { if constexpr (is_vec<V>) { bool r = false; for (int i = 0; i < V::dim; i++) r = r || v[i]; return r; } else return v; }
template <class T, int N> constexpr T w_dot(const vec<T, N>& a, const vec<T, N>& b) { T r = a[0] * b[0]; for (int i = 1; i < N; i++) r = r + a[i] * b[i]; return r; }
template <class V> constexpr auto w_length(const V& v) { if constexpr (is_vec<V>) return w_sqrt(w_dot(v, v)); else return w_abs(v); }
template <class A, class B> constexpr auto w_distance(const A& a, const B& b) { return w_length(a - b); }
template <class V> constexpr V w_normalize(const V& v) { return w_div(v, w_length(v)); }

// pack2x16unorm / pack4x8unorm: component i -> floor(0.5 + MAX * clamp(e_i, 0, 1)) in bits [i*W, i*W + W)
inline u32 unorm_bits(f32 e, f32 max) { f32 c = e < 0.0f ? 0.0f : (e > 1.0f ? 1.0f : e); if (e != e) c = 0.0f; return u32(std::floor(0.5f + max * c)); }
inline u32 w_pack2x16unorm(const vec2<f32>& e) { return unorm_bits(e.x, 65535.0f) | (unorm_bits(e.y, 65535.0f) << 16); }
inline u32 w_pack4x8unorm(const vec4<f32>& e) { return unorm_bits(e.x, 255.0f) | (unorm_bits(e.y, 255.0f) << 8) | (unorm_bits(e.z, 255.0f) << 16) | (unorm_bits(e.w, 255.0f) << 24); }

}  // namespace wgsl
