'use strict';
/*
 * ic.js -- initial conditions for the JavaScript host, in the reference's
 * packed layout: returns [bodies, vel], both Float32Array(4*N),
 * bodies = [x,y,z,m, ...], vel = [vx,vy,vz,0, ...] (/root/reference nbody3d.js:132).
 *
 *   galaxies(list, opts)     seedable re-implementation of the reference's only
 *                            generator, generateGalaxy (nbody3d.js:51-133):
 *                            rotating disks around a 1e7 central mass
 *   galaxySettings(k, opts)  the random galaxy list main() draws (nbody3d.js:167-175)
 *   plummer(n, opts)         Plummer sphere (BASELINE.json configs 1, 3, 5)
 *   uniformCube(n, opts)     uniform cube at rest (BASELINE.json config 2)
 *
 * `opts.random` is any () => [0,1) function (default: mulberry32(opts.seed)).
 * galaxies() consumes random numbers in the reference's order and rounds to
 * binary32 where the reference's Float32Array-backed vec3 helpers do, so that
 * with the same stream it returns the same bits (tests/js/node_tests.js checks
 * this against a fixture produced by running the reference's generator text).
 */
const f = Math.fround;

function mulberry32(seed) {
  let a = seed | 0;
  return function () {
    a = (a + 0x6D2B79F5) | 0;
    let t = Math.imul(a ^ (a >>> 15), 1 | a);
    t = (t + Math.imul(t ^ (t >>> 7), 61 | t)) ^ t;
    return ((t ^ (t >>> 14)) >>> 0) / 4294967296;
  };
}

function rngOf(opts) {
  if (opts && typeof opts.random === 'function') return opts.random;
  return mulberry32(opts && opts.seed !== undefined ? opts.seed : 1);
}

/* f32-rounded 3-vectors: the reference keeps every intermediate vector in a
 * Float32Array(3) (matrix.js:6-60), i.e. each component is rounded on store. */
function v3(x, y, z) { return [f(x), f(y), f(z)]; }
function vScale(a, s) { return v3(a[0] * s, a[1] * s, a[2] * s); }
function vAdd(a, b) { return v3(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
function vCross(a, b) { return v3(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]); }
function vLen(a) { return Math.sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
function vUnit(a) { const l = vLen(a); return l > 0 ? vScale(a, 1 / l) : v3(0, 0, 0); }

const CENTRAL_MASS = 1e7, OUTER_MASS_MAX = 50, OUTER_MASS_MIN = 10;   // nbody3d.js:62-64
const sphereRadius = function (mass) { return Math.cbrt(mass / (4 / 3 * Math.PI)); };   // util.js:123 (density 1)

/**
 * list: [[center[3], centerVelocity[3], normal[3], radius, count], ...]
 * opts: {G = 1e-4 (nbody3d.js:6), sizeFactor = 1080 (window.outerHeight, nbody3d.js:9), random | seed}
 */
function galaxies(list, opts) {
  const o = opts || {};
  const rand = rngOf(o);
  const G = o.G !== undefined ? o.G : 1e-4;
  const sizeFactor = o.sizeFactor !== undefined ? o.sizeFactor : 1080;
  const uniform = function (lo, hi) { return rand() * (hi - lo) + lo; };   // util.js:131
  let total = 0;
  for (let k = 0; k < list.length; k++) total += list[k][4] + 1;           // nbody3d.js:60
  const bodies = new Float32Array(4 * total), vel = new Float32Array(4 * total);
  const coreRadius = (sphereRadius(CENTRAL_MASS) + sphereRadius(OUTER_MASS_MAX)) / sizeFactor;   // :65
  let at = 0;
  for (let k = 0; k < list.length; k++) {
    const centre = list[k][0], drift = list[k][1], radius = list[k][3], count = list[k][4];
    // the central body (:66-68)
    bodies.set([centre[0], centre[1], centre[2], CENTRAL_MASS], 4 * at);
    vel.set([drift[0], drift[1], drift[2], 0], 4 * at);
    at++;
    // orthonormal frame of the disk plane (:75-84)
    const nrm = vUnit(list[k][2]);
    const helper = Math.abs(nrm[0]) > 0.9 ? v3(0, 1, 0) : v3(1, 0, 0);
    const e1 = vUnit(vCross(helper, nrm));
    const e2 = vCross(nrm, e1);
    for (let i = 0; i < count; i++) {
      // four draws per body, in this order: mass, radial parameter, angle, thickness (:88-100)
      const mass = uniform(OUTER_MASS_MIN, OUTER_MASS_MAX);
      const t = Math.sqrt(rand());
      const r = coreRadius + radius * (Math.pow(2, -2 * (t - 1)) - 1) / (Math.pow(2, 2) - 1);   // :93, exp = 2
      const theta = uniform(0, 2 * Math.PI);
      const lift = vScale(nrm, uniform(-0.1, 0.1) * (1 / (10 * Math.pow(r / radius, 2) + 1)));  // :100
      const inPlane = Math.sqrt(r * r - Math.pow(vLen(lift), 2));                                // :103-104
      const along1 = vScale(e1, inPlane * Math.cos(theta));
      const along2 = vScale(e2, inPlane * Math.sin(theta));
      const p = vAdd(vAdd(centre, lift), vAdd(along1, along2));                                  // :107
      bodies.set([p[0], p[1], p[2], mass], 4 * at);
      // circular speed about the central mass, tangent direction theta + pi/2 (:114-123)
      const speed = Math.sqrt(G * CENTRAL_MASS / r);
      const tang = theta + Math.PI / 2;
      const w = vAdd(drift, vAdd(vScale(e1, speed * Math.cos(tang)), vScale(e2, speed * Math.sin(tang))));
      vel.set([w[0], w[1], w[2], 0], 4 * at);
      at++;
    }
  }
  return [bodies, vel];
}

/** nbody3d.js:167-175 with the UI inputs as options (defaults index.html:68-74). */
function galaxySettings(numGalaxies, opts) {
  const o = opts || {};
  const rand = rngOf(o);
  const uniform = function (lo, hi) { return rand() * (hi - lo) + lo; };
  const minBodies = o.minBodies !== undefined ? o.minBodies : 20000;
  const maxBodies = o.maxBodies !== undefined ? o.maxBodies : 20000;
  const list = [];
  for (let i = 0; i < numGalaxies; i++) {
    list.push([
      [uniform(-5, 5), uniform(-5, 5), uniform(-5, 5)],
      [uniform(-10, 10), uniform(-10, 10), uniform(-10, 10)],
      [rand(), rand(), rand()],
      uniform(2, 5),
      Math.round(uniform(minBodies, maxBodies)),
    ]);
  }
  return list;
}

function isotropic(rand) {
  const z = 2 * rand() - 1, phi = 2 * Math.PI * rand(), s = Math.sqrt(1 - z * z);
  return [s * Math.cos(phi), s * Math.sin(phi), z];
}

/** Plummer sphere, Aarseth-Henon-Wielen sampling, N-body units (M = G = 1, virial radius 1). */
function plummer(n, opts) {
  const rand = rngOf(opts), rcut = (opts && opts.rcut) || 10;
  const a = 3 * Math.PI / 16;
  const P = new Float64Array(3 * n), V = new Float64Array(3 * n);
  for (let i = 0; i < n; i++) {
    let r;
    do { const x = Math.max(rand(), 1e-10); r = 1 / Math.sqrt(Math.pow(x, -2 / 3) - 1); } while (r > rcut);
    const d = isotropic(rand);
    let q, g;
    do { q = rand(); g = 0.1 * rand(); } while (g >= q * q * Math.pow(1 - q * q, 3.5));
    const vmag = q * Math.SQRT2 * Math.pow(1 + r * r, -0.25);
    const e = isotropic(rand);
    for (let c = 0; c < 3; c++) { P[3 * i + c] = a * r * d[c]; V[3 * i + c] = vmag * e[c] / Math.sqrt(a); }
  }
  const mp = [0, 0, 0], mv = [0, 0, 0];
  for (let i = 0; i < n; i++) for (let c = 0; c < 3; c++) { mp[c] += P[3 * i + c] / n; mv[c] += V[3 * i + c] / n; }
  const bodies = new Float32Array(4 * n), vel = new Float32Array(4 * n);
  for (let i = 0; i < n; i++) {
    for (let c = 0; c < 3; c++) { bodies[4 * i + c] = P[3 * i + c] - mp[c]; vel[4 * i + c] = V[3 * i + c] - mv[c]; }
    bodies[4 * i + 3] = 1 / n;
  }
  return [bodies, vel];
}

/** Positions uniform in [-1,1)^3, masses uniform in [0.5,1.5)/N, at rest. */
function uniformCube(n, opts) {
  const rand = rngOf(opts);
  const bodies = new Float32Array(4 * n), vel = new Float32Array(4 * n);
  for (let i = 0; i < n; i++) {
    bodies[4 * i] = 2 * rand() - 1; bodies[4 * i + 1] = 2 * rand() - 1; bodies[4 * i + 2] = 2 * rand() - 1;
    bodies[4 * i + 3] = (0.5 + rand()) / n;
  }
  return [bodies, vel];
}

module.exports = { galaxies: galaxies, galaxySettings: galaxySettings, plummer: plummer, uniformCube: uniformCube, mulberry32: mulberry32 };
