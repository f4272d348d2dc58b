// pbbi_autodiff.h -- forward-mode automatic differentiation for user-defined potentials.
//
// The reference's default path is gradient = jax.grad(potential) (src/HMC.py:57-60).  Here a user
// source (custom.py) that states only `potential` gets its gradient from DUAL NUMBERS: the same
// `potential` template is instantiated on a chain view whose elements are Dual{value, derivative}
// with the derivative of coordinate k seeded to 1; the derivative part of the result is dU/dq_k.
// D passes per gradient, each about twice a potential evaluation: meant for small D, or for
// checking a hand-written gradient (which stays the fast path).
//
// This text is pasted at the top of `namespace user` of every generated translation unit, device
// (hipcc) and host (the test oracle's g++ build) alike: no #include, PBBI_FN and T come from the
// surrounding unit.  Potentials that are to be differentiated must be written on the element type
// of their view -- `pbbi_scalar<Q> s = 0;` instead of `T s = 0;` -- and call math functions
// unqualified (argument-dependent lookup finds the overloads below).
#ifdef __HIPCC__
#define PBBI_AD_MEMBER __device__ __forceinline__
#else
#define PBBI_AD_MEMBER inline  // PBBI_FN is `static inline` in the host build: not for members
#endif
namespace pbbi_ad {

struct Dual {
    T v, d;
    PBBI_AD_MEMBER Dual() : v(0), d(0) {}
    PBBI_AD_MEMBER Dual(T value) : v(value), d(0) {}
    PBBI_AD_MEMBER Dual(T value, T deriv) : v(value), d(deriv) {}
    PBBI_AD_MEMBER Dual& operator+=(const Dual& o) { v += o.v; d += o.d; return *this; }
    PBBI_AD_MEMBER Dual& operator-=(const Dual& o) { v -= o.v; d -= o.d; return *this; }
    PBBI_AD_MEMBER Dual& operator*=(const Dual& o) { d = d * o.v + v * o.d; v *= o.v; return *this; }
    PBBI_AD_MEMBER Dual& operator/=(const Dual& o) { d = (d - (v / o.v) * o.d) / o.v; v /= o.v; return *this; }
};

PBBI_FN Dual operator+(const Dual& a, const Dual& b) { return Dual(a.v + b.v, a.d + b.d); }
PBBI_FN Dual operator-(const Dual& a, const Dual& b) { return Dual(a.v - b.v, a.d - b.d); }
PBBI_FN Dual operator*(const Dual& a, const Dual& b) { return Dual(a.v * b.v, a.d * b.v + a.v * b.d); }
PBBI_FN Dual operator/(const Dual& a, const Dual& b) {
    const T q = a.v / b.v;
    return Dual(q, (a.d - q * b.d) / b.v);
}
PBBI_FN Dual operator-(const Dual& a) { return Dual(-a.v, -a.d); }
PBBI_FN Dual operator+(const Dual& a) { return a; }
// mixed with plain scalars (any arithmetic type converts to T first)
#define PBBI_AD_MIXED(OP)                                                             \
    template <class S, class = std::enable_if_t<std::is_arithmetic<S>::value>>        \
    PBBI_FN Dual operator OP(const Dual& a, S b) { return a OP Dual((T)b); }          \
    template <class S, class = std::enable_if_t<std::is_arithmetic<S>::value>>        \
    PBBI_FN Dual operator OP(S a, const Dual& b) { return Dual((T)a) OP b; }
PBBI_AD_MIXED(+)
PBBI_AD_MIXED(-)
PBBI_AD_MIXED(*)
PBBI_AD_MIXED(/)
#undef PBBI_AD_MIXED
// comparisons look at the value only
#define PBBI_AD_CMP(OP)                                                               \
    PBBI_FN bool operator OP(const Dual& a, const Dual& b) { return a.v OP b.v; }     \
    template <class S, class = std::enable_if_t<std::is_arithmetic<S>::value>>        \
    PBBI_FN bool operator OP(const Dual& a, S b) { return a.v OP (T)b; }              \
    template <class S, class = std::enable_if_t<std::is_arithmetic<S>::value>>        \
    PBBI_FN bool operator OP(S a, const Dual& b) { return (T)a OP b.v; }
PBBI_AD_CMP(<)
PBBI_AD_CMP(>)
PBBI_AD_CMP(<=)
PBBI_AD_CMP(>=)
PBBI_AD_CMP(==)
PBBI_AD_CMP(!=)
#undef PBBI_AD_CMP

// elementary functions: value by the scalar function, derivative by the chain rule
PBBI_FN Dual exp(const Dual& a) { const T e = ::exp(a.v); return Dual(e, e * a.d); }
PBBI_FN Dual expm1(const Dual& a) { const T e = ::expm1(a.v); return Dual(e, (e + T(1)) * a.d); }
PBBI_FN Dual log(const Dual& a) { return Dual(::log(a.v), a.d / a.v); }
PBBI_FN Dual log1p(const Dual& a) { return Dual(::log1p(a.v), a.d / (T(1) + a.v)); }
PBBI_FN Dual sqrt(const Dual& a) { const T r = ::sqrt(a.v); return Dual(r, a.d / (T(2) * r)); }
PBBI_FN Dual sin(const Dual& a) { return Dual(::sin(a.v), ::cos(a.v) * a.d); }
PBBI_FN Dual cos(const Dual& a) { return Dual(::cos(a.v), -::sin(a.v) * a.d); }
PBBI_FN Dual tanh(const Dual& a) { const T t = ::tanh(a.v); return Dual(t, (T(1) - t * t) * a.d); }
PBBI_FN Dual atan(const Dual& a) { return Dual(::atan(a.v), a.d / (T(1) + a.v * a.v)); }
PBBI_FN Dual fabs(const Dual& a) { return a.v < T(0) ? -a : a; }
PBBI_FN Dual fma(const Dual& a, const Dual& b, const Dual& c) { return a * b + c; }
template <class S, class = std::enable_if_t<std::is_arithmetic<S>::value>>
PBBI_FN Dual pow(const Dual& a, S e) {  // constant exponent
    const T p = ::pow(a.v, (T)e - T(1));
    return Dual(p * a.v, (T)e * p * a.d);
}
PBBI_FN Dual fmax(const Dual& a, const Dual& b) { return a.v < b.v ? b : a; }
PBBI_FN Dual fmin(const Dual& a, const Dual& b) { return b.v < a.v ? b : a; }

// the chain with coordinate k seeded: view[j] = Dual{q[j], j == k}
template <class Q>
struct Seeded {
    const Q& q;
    int k;
    PBBI_AD_MEMBER Dual operator[](int j) const { return Dual((T)q[j], j == k ? T(1) : T(0)); }
};

}  // namespace pbbi_ad

// element type of a chain view: T for the kernels' views, pbbi_ad::Dual under differentiation
// (no std::declval: it is a host function to hipcc's device pass)
template <class Q>
using pbbi_scalar = std::decay_t<decltype((*static_cast<const Q*>(nullptr))[0])>;
