"""Derive the separable form of the Matern-2.5 linked-GP J factor (vecchia.py:915-959).

Jd(x1,x2) = P1+P2+P3 where every erf/exp depends on ONE of the two points.  Writing each E-polynomial as
sum_a other^a * e_a(x) gives  Jd = <S(x_min), T(x_max)>  with 12-component vectors (tools output: the e_a).
Prints C expressions for the coefficient polynomials e{3,4,5}{0..4}_a(x) (a = power of the OTHER point)."""
import sympy as sp

x1, x2, l = sp.symbols('x1 x2 l')
s5 = sp.sqrt(5)
l4 = 9 * l**4
E = {}
E['30'] = 1 + (25*x1**2*x2**2 - 3*s5*(3*l**3 + 5*l*x1*x2)*(x1+x2) + 15*l**2*(x1**2 + x2**2 + 3*x1*x2)) / l4
E['31'] = (18*s5*l**3 + 15*s5*l*(x1**2+x2**2) - (75*l**2 + 50*x1*x2)*(x1+x2) + 60*s5*l*x1*x2) / l4
E['32'] = 5*(5*x1**2 + 5*x2**2 + 15*l**2 - 9*s5*l*(x1+x2) + 20*x1*x2) / l4
E['33'] = 10*(3*s5*l - 5*x1 - 5*x2) / l4
E['34'] = sp.Integer(25) / l4
E['40'] = 1 + (25*x1**2*x2**2 + 3*s5*(3*l**3 - 5*l*x1*x2)*(x2-x1) + 15*l**2*(x1**2 + x2**2 - 3*x1*x2)) / l4
E['41'] = 5*(3*s5*l*(x2**2 - x1**2) + 3*l**2*(x1+x2) - 10*x1*x2*(x1+x2)) / l4
E['42'] = 5*(5*x1**2 + 5*x2**2 - 3*l**2 - 3*s5*l*(x2-x1) + 20*x1*x2) / l4
E['43'] = -50*(x1+x2) / l4
E['44'] = sp.Integer(25) / l4
E['50'] = 1 + (25*x1**2*x2**2 + 3*s5*(3*l**3 + 5*l*x1*x2)*(x1+x2) + 15*l**2*(x1**2 + x2**2 + 3*x1*x2)) / l4
E['51'] = (18*s5*l**3 + 15*s5*l*(x1**2+x2**2) + (75*l**2 + 50*x1*x2)*(x1+x2) + 60*s5*l*x1*x2) / l4
E['52'] = 5*(5*x1**2 + 5*x2**2 + 15*l**2 + 9*s5*l*(x1+x2) + 20*x1*x2) / l4
E['53'] = 10*(3*s5*l + 5*x1 + 5*x2) / l4
E['54'] = sp.Integer(25) / l4

x = sp.Symbol('x')


def coeffs(expr, other, keep):
    """coefficients of other^a (a=0,1,2) as polynomials in `keep` renamed to x, times 9 l^4 (common factor)."""
    p = sp.Poly(sp.expand(expr * l4), other)
    assert p.degree() <= 2, (expr, p.degree())
    out = []
    for a in range(3):
        c = p.coeff_monomial(other**a)
        out.append(sp.horner(sp.expand(c.subs(keep, x)), wrt=x))
    return out


def cc(e):
    s = sp.ccode(e)
    return s.replace('sqrt(5)', 'SQRT5').replace('pow(l, 2)', 'l2').replace('pow(l, 3)', 'l3').replace('pow(x, 2)', '(x*x)')


print('// coefficient of OTHER^a (a=0..2) in 9 l^4 * E_ij, as a polynomial in the point x itself')
print('// "L": x is the LARGER point x2, OTHER = x1 (E3*, E4*) ; "S": x is the SMALLER point x1, OTHER = x2 (E5*, E4*)')
for tag, other, keep, names in (('L', x1, x2, ['30', '31', '32', '33', '34', '40', '41', '42', '43', '44']),
                                ('S', x2, x1, ['50', '51', '52', '53', '54', '40', '41', '42', '43', '44'])):
    for nm in names:
        cs = coeffs(E[nm], other, keep)
        print('// %s e%s' % (tag, nm))
        for a, c in enumerate(cs):
            print('const double %s%s_%d = %s;' % (tag.lower(), nm, a, cc(c)))
