'use strict';
/*
 * js_oracle.js -- single-thread JavaScript restatement of the reference's
 * force + integrate pass, binary32 via Math.fround after every operation.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/nb_oracle.c header): used by the Node
 * parity tests and as the "JS without a GPU" timing figure of BASELINE.md §4.
 * It must produce the SAME BITS as oracle/nb_oracle.c's f32 path (checked
 * against the committed golden vectors in tests/js/node_tests.js).
 *
 * Follows /root/reference nbody3d.js:232-237 (pair force), :255-272 (ascending
 * j, j != i), :274-290 (integrator), with synchronous semantics.
 * The two WGSL fma() calls need a single rounding: fmaF32() emulates fmaf with
 * doubles (exact product of two f32 fits a double; the double add can double
 * round in rare ties, detected and corrected below).
 */
const f = Math.fround;

/* fmaf(a,b,c) for f32 inputs.  a*b is exact in double (24+24 bits).  The sum
 * p + c is rounded to double then to float: a double-rounding error is only
 * possible when the double result sits exactly on a float rounding boundary;
 * in that case redo the addition with an error-free transformation. */
function fmaF32(a, b, c) {
  const p = a * b;            // exact
  const s = p + c;            // rounded to double
  const r = f(s);
  // fast path: s is not a tie candidate
  const err = (s - p) - c;    // == -(rounding error of p + c) when |p| >= |c| or vice versa (two-sum, simplified)
  if (err === 0) return r;    // the double add was exact -> single rounding
  // slow path: exact two-sum and sticky correction
  const bb = s - p;
  const e = (p - (s - bb)) + (c - bb);   // s + e == p + c exactly
  if (e === 0) return r;
  // is s exactly halfway between two adjacent floats?
  const lo = f(s);
  if (lo === s) {
    // s is representable as float but the true sum is s + e: direction matters only
    // if rounding (s+e) to float differs -- it cannot, |e| < ulp_double(s)/2 << ulp_float(s)/2
    return lo;
  }
  const up = lo < s ? nextUp(lo) : lo;
  const dn = lo < s ? lo : nextDown(lo);
  const mid = (up + dn) / 2;  // exact in double
  if (s !== mid) return r;    // not a tie in double -> r is already correct
  return e > 0 ? up : dn;     // true sum is just above / below the tie
}
const _fb = new Float32Array(1), _ib = new Int32Array(_fb.buffer);
function nextUp(x) { _fb[0] = x; if (x >= 0) _ib[0] += 1; else _ib[0] -= 1; return _fb[0]; }
function nextDown(x) { _fb[0] = x; if (x > 0) _ib[0] -= 1; else if (x < 0) _ib[0] += 1; else return -1.401298464324817e-45; return _fb[0]; }

function accelF32(bodies, n, G, eps2, out) {
  G = f(G); eps2 = f(eps2);
  for (let i = 0; i < n; i++) {
    const xi = bodies[4 * i], yi = bodies[4 * i + 1], zi = bodies[4 * i + 2];
    let ax = 0, ay = 0, az = 0;
    for (let j = 0; j < n; j++) {
      if (j === i) continue;                                           // nbody3d.js:265
      const rx = f(bodies[4 * j] - xi), ry = f(bodies[4 * j + 1] - yi), rz = f(bodies[4 * j + 2] - zi);
      const d2 = f(f(f(f(rx * rx) + f(ry * ry)) + f(rz * rz)) + eps2);  // :234
      const d6 = f(f(d2 * d2) * d2);                                   // :235
      const inv = f(1 / f(Math.sqrt(d6)));
      const s = f(f(G * bodies[4 * j + 3]) * inv);                     // :236
      ax = f(ax + f(s * rx)); ay = f(ay + f(s * ry)); az = f(az + f(s * rz));
    }
    out[4 * i] = ax; out[4 * i + 1] = ay; out[4 * i + 2] = az; out[4 * i + 3] = 0;
  }
}

/* state arrays are Float32Array(4n), updated in place */
function stepF32(bodies, vel, accel, n, dt, G, eps2, scratch) {
  if (!(dt > 0)) return;                                               // :474
  dt = f(dt);
  accelF32(bodies, n, G, eps2, scratch);
  const h = f(dt * 0.5);                                               // :276
  for (let k = 0; k < 4 * n; k++) {
    const nv = fmaF32(f(accel[k] + scratch[k]), h, vel[k]);            // :280
    vel[k] = nv;
    bodies[k] = fmaF32(fmaF32(h, scratch[k], nv), dt, bodies[k]);      // :283
    accel[k] = scratch[k];                                             // :290
  }
}

function runF32(bodies0, vel0, accel0, dt, G, nsteps, eps2) {
  const n = bodies0.length / 4;
  const b = Float32Array.from(bodies0), v = Float32Array.from(vel0);
  const a = accel0 ? Float32Array.from(accel0) : new Float32Array(4 * n);
  const scratch = new Float32Array(4 * n);
  for (let s = 0; s < nsteps; s++) stepF32(b, v, a, n, dt, G, eps2 === undefined ? 1e-4 : eps2, scratch);
  return { bodies: b, vel: v, accel: a };
}

module.exports = { accelF32: accelF32, stepF32: stepF32, runF32: runF32, fmaF32: fmaF32 };
