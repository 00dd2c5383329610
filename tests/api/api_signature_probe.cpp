// tests/api/api_signature_probe.cpp -- the EXACT types of the public members of dj_brdf.h:54-537, asserted at compile time.  The file
// compiles against the reference (which proves the assertions) and must compile against include/dj_brdf.h: a member whose return type,
// parameter list, const-ness or static-ness differs is a compile error here, where tests/api/api_surface_probe.cpp (which only CALLS
// every name once) would accept a compatible-looking overload.  Overloaded members are selected by casting to the expected type.
// Default arguments are exercised by calls with the trailing arguments omitted.   (C++11: decltype / static_assert)
#include <type_traits>
#include <vector>
#define DJ_BRDF_IMPLEMENTATION 1
#include "dj_brdf.h"

using djb::vec3;
typedef djb::float_t F;
typedef djb::microfacet::params P;
#define SAME(expr, ...) static_assert(std::is_same<decltype(expr), __VA_ARGS__>::value, #expr)
// ---- vec3, exc
SAME(&vec3::intensity, F (vec3::*)() const);
SAME(static_cast<vec3 (*)(const double *)>(&vec3::from_raw), vec3 (*)(const double *));
SAME(static_cast<vec3 (*)(const float *)>(&vec3::from_raw), vec3 (*)(const float *));
SAME(&vec3::to_raw, const F *(*)(const vec3 &));
SAME(&djb::exc::what, const char *(djb::exc::*)() const noexcept);
SAME(&djb::exc::m_str, std::string djb::exc::*);
static_assert(std::is_base_of<std::exception, djb::exc>::value, "exc derives from std::exception");
// ---- brdf
SAME(static_cast<vec3 (djb::brdf::*)(const vec3 &, const vec3 &, const void *) const>(&djb::brdf::eval), vec3 (djb::brdf::*)(const vec3 &, const vec3 &, const void *) const);
SAME(static_cast<vec3 (djb::brdf::*)(const vec3 &, const vec3 &, const void *) const>(&djb::brdf::eval_hd), vec3 (djb::brdf::*)(const vec3 &, const vec3 &, const void *) const);
SAME(static_cast<vec3 (djb::brdf::*)(const vec3 &, const vec3 &, const void *) const>(&djb::brdf::evalp), vec3 (djb::brdf::*)(const vec3 &, const vec3 &, const void *) const);
SAME(static_cast<vec3 (djb::brdf::*)(const vec3 &, const vec3 &, const void *) const>(&djb::brdf::evalp_hd), vec3 (djb::brdf::*)(const vec3 &, const vec3 &, const void *) const);
SAME(static_cast<vec3 (djb::brdf::*)(F, F, const vec3 &, vec3 *, F *, const void *) const>(&djb::brdf::evalp_is), vec3 (djb::brdf::*)(F, F, const vec3 &, vec3 *, F *, const void *) const);
SAME(static_cast<vec3 (djb::brdf::*)(F, F, const vec3 &, const void *) const>(&djb::brdf::sample), vec3 (djb::brdf::*)(F, F, const vec3 &, const void *) const);
SAME(static_cast<F (djb::brdf::*)(const vec3 &, const vec3 &, const void *) const>(&djb::brdf::pdf), F (djb::brdf::*)(const vec3 &, const vec3 &, const void *) const);
SAME(&djb::brdf::io_to_hd, void (*)(const vec3 &, const vec3 &, vec3 *, vec3 *));
SAME(&djb::brdf::hd_to_io, void (*)(const vec3 &, const vec3 &, vec3 *, vec3 *));
static_assert(std::is_abstract<djb::brdf>::value && std::has_virtual_destructor<djb::brdf>::value, "brdf: abstract, virtual destructor");
static_assert(!std::is_copy_constructible<djb::brdf>::value && !std::is_copy_assignable<djb::ggx>::value, "noncopyable");
// (djb::microfacet is abstract in the reference as well; the facade's has defaults that throw "Not Implemented" instead of pure virtuals,
// because the library's own lobes are answered from HBM and implement none of them: more permissive, never less)
static_assert(std::is_abstract<djb::fresnel::impl>::value && std::is_abstract<djb::radial>::value, "the extension points are abstract");
SAME(&djb::lambert::params::m_reflectance, vec3 djb::lambert::params::*);
// ---- merl / utia
SAME(&djb::merl::get_samples, const std::vector<double> &(djb::merl::*)() const);
SAME(&djb::utia::get_samples, const std::vector<double> &(djb::utia::*)() const);
static_assert(std::is_constructible<djb::merl, const char *>::value && std::is_constructible<djb::utia, const char *>::value, "file constructors");
// ---- fresnel
SAME(&djb::fresnel::impl::eval, vec3 (djb::fresnel::impl::*)(F) const);
SAME(&djb::fresnel::impl::copy, djb::fresnel::impl *(djb::fresnel::impl::*)() const);
SAME(&djb::fresnel::spline::get_points, const std::vector<vec3> &(djb::fresnel::spline::*)() const);
SAME(static_cast<void (*)(F, F *)>(&djb::fresnel::ior_to_f0), void (*)(F, F *));
SAME(static_cast<void (*)(const vec3 &, vec3 *)>(&djb::fresnel::f0_to_ior), void (*)(const vec3 &, vec3 *));
static_assert(std::is_constructible<djb::fresnel::sgd, const vec3 &, const vec3 &>::value && std::is_convertible<vec3, djb::fresnel::schlick>::value
              && std::is_convertible<vec3, djb::fresnel::unpolarized>::value && !std::is_convertible<std::vector<vec3>, djb::fresnel::spline>::value, "Fresnel constructors (spline's is explicit)");
// ---- microfacet::params
SAME(&P::standard, P (*)());
SAME(&P::isotropic, P (*)(F));
SAME(&P::elliptic, P (*)(F, F, F));
SAME(&P::pdfparams, P (*)(F, F, F, F, F));
SAME(&P::set_ellipse, void (P::*)(F, F, F));
SAME(&P::set_pdfparams, void (P::*)(F, F, F, F, F));
SAME(static_cast<void (P::*)(F, F)>(&P::set_location), void (P::*)(F, F));
SAME(static_cast<void (P::*)(const vec3 &)>(&P::set_location), void (P::*)(const vec3 &));
SAME(&P::get_ellipse, void (P::*)(F *, F *, F *) const);
SAME(&P::get_pdfparams, void (P::*)(F *, F *, F *, F *, F *) const);
SAME(static_cast<void (P::*)(F *, F *) const>(&P::get_location), void (P::*)(F *, F *) const);
SAME(static_cast<void (P::*)(vec3 *) const>(&P::get_location), void (P::*)(vec3 *) const);
// ---- microfacet
typedef djb::microfacet M;
SAME(&M::fresnel, vec3 (M::*)(F) const);
SAME(&M::ndf, F (M::*)(const vec3 &, const P &) const);
SAME(&M::gaf, F (M::*)(const vec3 &, const vec3 &, const vec3 &, const P &) const);
SAME(&M::g1, F (M::*)(const vec3 &, const vec3 &, const P &) const);
SAME(&M::sigma, F (M::*)(const vec3 &, const P &) const);
SAME(&M::p22, F (M::*)(F, F, const P &) const);
SAME(&M::vp22, F (M::*)(F, F, const vec3 &, const P &) const);
SAME(&M::vndf, F (M::*)(const vec3 &, const vec3 &, const P &) const);
SAME(&M::supports_smith_vndf_sampling, bool (M::*)() const);
SAME(&M::qf2, F (M::*)(F, const vec3 &) const);
SAME(&M::qf3, F (M::*)(F, const vec3 &, F) const);
SAME(&M::set_shadow, void (M::*)(bool));
SAME(&M::set_fresnel, void (M::*)(const djb::fresnel::impl &));
SAME(&M::get_shadow, int (M::*)() const);
SAME(&M::get_fresnel, const djb::fresnel::impl &(M::*)() const);
// ---- radial and the three radial lobes
typedef djb::radial R;
SAME(&R::p22_radial, F (R::*)(F) const);
SAME(&R::sigma_std_radial, F (R::*)(F) const);
SAME(&R::cdf_radial, F (R::*)(F) const);
SAME(&R::qf_radial, F (R::*)(F) const);
SAME(&R::qf2_radial, F (R::*)(F, F, F) const);
SAME(&R::qf3_radial, F (R::*)(F, F) const);
SAME(&djb::beckmann::qf1, F (djb::beckmann::*)(F) const);
SAME(&djb::ggx::qf1, F (djb::ggx::*)(F) const);
SAME(&djb::beckmann::params_to_lrep, void (*)(const P &, djb::beckmann::lrep *));
SAME(&djb::beckmann::lrep_to_params, void (*)(const djb::beckmann::lrep &, P *));
typedef djb::beckmann::lrep L;
SAME(&L::operator+, L (L::*)(const L &) const);
SAME(&L::operator*, L (L::*)(F) const);
SAME(&L::operator+=, L &(L::*)(const L &));
SAME(&L::operator*=, L &(L::*)(F));
SAME(static_cast<void (L::*)(F, F)>(&L::scale), void (L::*)(F, F));
SAME(&L::shear, void (L::*)(F, F));
static_assert(std::is_constructible<L, F, F, F, F, F>::value && std::is_default_constructible<L>::value, "lrep constructors");
// ---- tabular / tabular_anisotropic
typedef djb::tabular T;
typedef djb::tabular_anisotropic A;
static_assert(std::is_constructible<T, const djb::brdf &, int>::value && std::is_constructible<T, const djb::brdf &, int, bool>::value, "tabular(brdf, res[, shadow])");
static_assert(std::is_constructible<A, const djb::brdf &, int, int>::value && std::is_constructible<A, const djb::brdf &, int, int, bool>::value, "tabular_anisotropic(brdf, e, a[, shadow])");
SAME(&T::fit_beckmann_parameters, P (*)(const T &));
SAME(&T::fit_ggx_parameters, P (*)(const T &));
SAME(&T::get_p22v, const std::vector<F> &(T::*)() const);
SAME(&T::get_sigmav, const std::vector<F> &(T::*)() const);
SAME(&T::get_cdfv, const std::vector<F> &(T::*)() const);
SAME(&T::get_qfv, const std::vector<F> &(T::*)() const);
SAME(&A::fit_beckmann_parameters, P (*)(const A &));
SAME(&A::fit_ggx_parameters, P (*)(const A &));
SAME(&A::get_p22v, const std::vector<F> &(A::*)(int *, int *) const);
SAME(&A::get_sigmav, const std::vector<F> &(A::*)(int *, int *) const);
SAME(&A::pdf1, F (A::*)(F) const);
SAME(&A::cdf1, F (A::*)(F) const);
SAME(&A::qf1, F (A::*)(F) const);
SAME(&A::pdf2, F (A::*)(F, F) const);
SAME(&A::cdf2, F (A::*)(F, F) const);
SAME(&A::qf2, F (A::*)(F, F) const);
// ---- the published models
SAME(&djb::sgd::ndf, vec3 (djb::sgd::*)(const vec3 &) const);
SAME(&djb::sgd::gaf, vec3 (djb::sgd::*)(const vec3 &, const vec3 &, const vec3 &) const);
SAME(&djb::sgd::g1, vec3 (djb::sgd::*)(const vec3 &) const);
SAME(&djb::sgd::fresnel, vec3 (djb::sgd::*)(F) const);
SAME(&djb::sgd::get_fresnel, const djb::fresnel::impl &(djb::sgd::*)() const);
SAME(&djb::abc::ndf, vec3 (djb::abc::*)(const vec3 &) const);
SAME(&djb::abc::gaf, F (djb::abc::*)(const vec3 &, const vec3 &, const vec3 &) const);
SAME(&djb::abc::fresnel, vec3 (djb::abc::*)(F) const);
SAME(&djb::abc::get_fresnel, const djb::fresnel::impl &(djb::abc::*)() const);
static_assert(std::is_convertible<const char *, djb::sgd>::value == std::is_convertible<const char *, djb::abc>::value, "name constructors alike");

// ---- default arguments: every trailing argument the reference lets a caller omit
static void defaults()
{
	vec3 a(0, 0, 1), v0, v1(0.5f), v3_(1, 2, 3);
	(void)v0; (void)v1; (void)v3_;
	djb::ggx g; djb::beckmann b; djb::lambert l;
	djb::ggx g2(djb::fresnel::ideal()); djb::beckmann b2(djb::fresnel::ideal(), false);
	(void)g.eval(a, a); (void)g.evalp(a, a); (void)g.pdf(a, a); (void)g.sample(0.5f, 0.5f, a); (void)g.eval_hd(a, a); (void)g.evalp_hd(a, a);
	vec3 wi; F pdf; (void)g.evalp_is(0.5f, 0.5f, a, &wi, &pdf);
	(void)g.ndf(a); (void)g.gaf(a, a, a); (void)g.g1(a, a); (void)g.sigma(a); (void)g.p22(0, 0); (void)g.vp22(0, 0, a); (void)g.vndf(a, a);
	(void)P::elliptic(0.1f, 0.2f); (void)P::pdfparams(0.1f, 0.2f); (void)P::pdfparams(0.1f, 0.2f, 0.0f, 0.1f);
	P p; p.set_ellipse(0.1f, 0.2f); p.set_pdfparams(0.1f, 0.2f); F x, y; p.get_ellipse(&x, &y); p.get_pdfparams(&x, &y);
	P p1(0.5f), p2(0.5f, 0.6f);
	(void)p1; (void)p2;
	djb::lambert::params lp; (void)l.eval(a, a, &lp);
	djb::beckmann::lrep r0, r1(0.1f), r2(0.1f, 0.2f, 1.0f);
	(void)r0; (void)r1; (void)r2;
	djb::tabular t(g, 8); djb::tabular_anisotropic ta(g, 6, 8);
	(void)t; (void)ta;
}
int main() { (void)&defaults; return 0; }
