import numpy as np


class CosmologyError(Exception):
    pass


class BaseEngine(object):
    pass


class BaseSection(object):
    pass


class SyntheticPk(object):
    """P(k) = A (k/0.05)^ns T_BBKS(k/keq)^2 (1 + wiggle sin(k rs) exp(-(8k)^2)), times ``scale``."""

    def __init__(self, A=None, n_s=0.965, keq=0.015, rs=100., wiggle=0.05, scale=1.):
        if A is None:   # REFSTUB_PK_SCALE: amplitude knob of the synthetic spectrum (the one-loop fixtures want loop terms of realistic relative size)
            import os
            A = 2.5e4 * float(os.environ.get('REFSTUB_PK_SCALE', '1'))
        self.A, self.n_s, self.keq, self.rs, self.wiggle, self.scale = A, n_s, keq, rs, wiggle, scale
        self.k = np.geomspace(1e-4, 10., 1201)      # tabulation grid of an interpolator (read by the reference's turn-over finder, power_template.py:1205-1220)

    def clone(self, **kwargs):
        state = dict(A=self.A, n_s=self.n_s, keq=self.keq, rs=self.rs, wiggle=self.wiggle, scale=self.scale)
        state.update(kwargs)
        return SyntheticPk(**state)

    def __call__(self, k, **kwargs):
        k = np.asarray(k, dtype='f8')
        q = k / self.keq
        T = np.log(1. + 2.34 * q) / (2.34 * q) * (1. + 3.89 * q + (16.1 * q)**2 + (5.46 * q)**3 + (6.71 * q)**4)**(-0.25)
        return self.scale * self.A * (k / 0.05)**self.n_s * T**2 * (1. + self.wiggle * np.sin(k * self.rs) * np.exp(-(8. * k)**2))

    def sigma_r(self, r, **kwargs):
        k = np.logspace(-5, 2, 4000)
        x = k * r
        w = 3. * (np.sin(x) - x * np.cos(x)) / x**3
        integrand = k**3 * self(k) * w**2 / (2. * np.pi**2)
        lnk = np.log(k)
        return np.sqrt(np.sum((integrand[1:] + integrand[:-1]) / 2. * np.diff(lnk)))

    def to_1d(self, z=0., **kwargs):
        return self


class _PkInterp2D(object):

    def __init__(self, pk):
        self.pk = pk

    def to_1d(self, z=0., **kwargs):
        return self.pk


class _Fourier(object):

    def __init__(self, cosmo):
        self.cosmo = cosmo

    def sigma8_z(self, z, of='delta_cb', **kwargs):
        return {'delta_cb': 0.8, 'theta_cb': 0.64}[of if isinstance(of, str) else of[0]] * np.ones_like(np.asarray(z, dtype='f8'))

    def sigma_rz(self, r, z, of='delta_cb', **kwargs):
        return self.sigma8_z(z, of=of) * (8. / np.asarray(r))**0.5

    def pk_interpolator(self, of='delta_cb', **kwargs):
        of = of if isinstance(of, str) else of[0]
        return _PkInterp2D(SyntheticPk(scale={'delta_cb': 1., 'theta_cb': 0.64}[of], rs=self.cosmo.rs_drag, n_s=self.cosmo.n_s))


class Cosmology(object):

    rs_drag = 100.
    n_s = 0.965
    h = 0.7

    def __init__(self, **kwargs):
        self.params = dict(kwargs)

    def efunc(self, z):
        z = np.asarray(z, dtype='f8')
        return np.sqrt(0.3 * (1. + z)**3 + 0.7)

    def comoving_angular_distance(self, z):
        z = np.asarray(z, dtype='f8')
        zz = np.linspace(0., 1., 2001)[:, None] * z
        return 2997.92458 * np.trapezoid(1. / self.efunc(zz), zz, axis=0)

    def get_fourier(self, *args, **kwargs):
        return _Fourier(self)

    def get_primordial(self, *args, **kwargs):
        cosmo = self

        class _Primordial(object):

            def pk_interpolator(self, **kwargs):
                return lambda k: 2.1e-9 * (np.asarray(k) / 0.05)**(cosmo.n_s - 1.)   # dimensionless primordial spectrum A_s (k / k_pivot)^(n_s - 1)

        return _Primordial()

    def clone(self, **kwargs):
        return Cosmology(**{**self.params, **kwargs})

    def __getstate__(self):
        return {}

    @classmethod
    def from_state(cls, state):
        return cls()
