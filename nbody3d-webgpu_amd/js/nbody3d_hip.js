'use strict';
/*
 * nbody3d_hip.js -- Node.js host side of the MI355X direct N-body engine.
 *
 * Keeps the JavaScript surface of the reference's force+integrate path.  The
 * reference (huj31415/nbody3d-webgpu) has no exported functions: the path is a
 * region of one browser script.  The names below are the ones BASELINE.json's
 * north_star uses, mapped onto that region (file:line in /root/reference):
 *
 *   init(particles)        nbody3d.js:177-199  generate -> 3x createBuffer ->
 *                                              2x writeBuffer (accel left zero)
 *   step(dt)               nbody3d.js:470 (uniform upload) + :474-480 (compute
 *                          pass, gated by dt > 0) + :489-490 (submit)
 *   simulate(nSteps, dt)   nSteps frames of step() without the render pass
 *   read()                 util.js:163-178     the three readBuffer() copies
 *   restore(state)         util.js:230-244     importSimulation's buffer writes
 *   G / dt / pause()       nbody3d.js:6-7, util.js:36-64 (dt and G are mutable
 *                          between frames; pause saves dt and sets it to 0)
 *   requestFrame()/frame() nbody3d.js:408-415,482-487 what the render pass reads each frame
 *                          (bodies + speed), delivered asynchronously to a host-side viewer
 *
 * Data convention is the reference's: Float32Array packed [x,y,z,m, ...] and
 * [vx,vy,vz,0, ...] (nbody3d.js:49,132).  Node >= 12 syntax only.
 *
 * All compute happens in csrc/libnbody3d_hip.so through addon/nb_napi.node;
 * there is no JavaScript fallback -- a missing library or GPU throws.
 */
const path = require('path');

const DEFAULT_LIB = path.join(__dirname, '..', 'csrc', 'libnbody3d_hip.so');
const TILE_SIZE = 256;      // nbody3d.js:4
const DEFAULT_G = 0.0001;   // nbody3d.js:6
const DEFAULT_DT = 1e-4;    // nbody3d.js:7
const EPS2 = 1e-4;          // nbody3d.js:234

let addon = null;
let abi = 0;

function load(libPath) {
  if (!addon) addon = require('./addon/nb_napi.node');
  abi = addon.load(libPath || process.env.NBODY3D_HIP_LIB || DEFAULT_LIB);
  return abi;
}

function deviceCount() {
  load();
  return addon.deviceCount();
}

/** The launch plan the engine would build for {n, f64, variant, jsplit, flags, layerBudgetMiB, shardBegin, shardCount}
 *  (nb_plan_query): kernel form, j-partitions, and the symmetric pass's padded rows / partial-sum layers / bytes.  With nCU and
 *  clockHz given (an MI355X: 256, 2.4e9) it needs no GPU -- capacity planning; otherwise the current device's figures are used. */
function planQuery(options) {
  load();
  return addon.planQuery(options || {});
}

function asParticles(particles) {
  // generateGalaxy returns [pos, vel] (nbody3d.js:132); objects are accepted too
  let bodies, vel, accel = null;
  if (Array.isArray(particles)) {
    bodies = particles[0]; vel = particles[1]; accel = particles[2] || null;
  } else if (particles && typeof particles === 'object') {
    bodies = particles.bodies; vel = particles.vel; accel = particles.accel || null;
  }
  if (!bodies || !vel) throw new TypeError('init(particles): expected [bodies, vel] or {bodies, vel[, accel]}');
  return { bodies: bodies, vel: vel, accel: accel };
}

class Simulation {
  /** options: {G, dt, f64, eps2, device, shards, collective, shardBegin, shardCount, variant, jsplit, flags, layerBudgetMiB}
   *  layerBudgetMiB: nb_config.layer_budget_mib (device memory the symmetric pass may take for its partial sums; 0 = default).
   *  shards > 1: single-process multi-device (i-shards round-robin over the visible GPUs; all-gather
   *  of positions after every step through collective: 'peer' -- event-ordered device-to-device
   *  copies, the default -- or 'rccl' -- ncclCommInitAll + grouped in-place ncclAllGather;
   *  no reference analogue). */
  constructor(options) {
    const o = options || {};
    this.options = o;
    this.G = o.G !== undefined ? o.G : DEFAULT_G;
    this.dt = o.dt !== undefined ? o.dt : DEFAULT_DT;
    this.f64 = !!o.f64;
    this._oldDt = null;     // util.js:35
    this._h = null;
    this.nBodies = 0;       // nbody3d.js:14
  }

  get ArrayType() { return this.f64 ? Float64Array : Float32Array; }

  _coerce(a, name) {
    const T = this.ArrayType;
    if (!(a instanceof T)) {
      if (a && typeof a.length === 'number') a = T.from(a);   // plain arrays as in importSimulation (util.js:231)
      else throw new TypeError(name + ': expected ' + T.name);
    }
    if (a.length !== 4 * this.nBodies) throw new RangeError(name + ': expected ' + (4 * this.nBodies) + ' elements, got ' + a.length);
    return a;
  }

  /** nbody3d.js:177-199.  Re-initialising replaces the buffers (util.js:69-75 "Regenerate"). */
  init(particles) {
    load();
    const p = asParticles(particles);
    if (p.bodies.length % 4 !== 0 || p.bodies.length === 0) throw new RangeError('bodies must hold 4*n elements');
    const n = p.bodies.length / 4;
    if (this._h && n !== this.nBodies) this.destroy();
    this.nBodies = n;
    if (!this._h) {
      const o = this.options;
      this._h = addon.create({
        n: n, f64: this.f64 ? 1 : 0, eps2: o.eps2 !== undefined ? o.eps2 : EPS2,
        device: o.device !== undefined ? o.device : -1, shardBegin: o.shardBegin || 0, shardCount: o.shardCount || 0,
        variant: o.variant || 0, jsplit: o.jsplit || 0, tile: o.tile || 0, shards: o.shards || 0,
        flags: o.flags || 0, layerBudgetMiB: o.layerBudgetMiB || 0, collective: o.collective === 'rccl' ? 1 : 0,
      });
      this._frame = null;
    }
    addon.upload(this._h, this._coerce(p.bodies, 'bodies'), this._coerce(p.vel, 'vel'),
      p.accel ? this._coerce(p.accel, 'accel') : null);
    return this;
  }

  _need() { if (!this._h) throw new Error('simulation not initialised: call init(particles) first'); }

  /** One frame's compute pass.  dt <= 0 (paused) dispatches nothing (nbody3d.js:474). */
  step(dt) {
    this._need();
    if (dt !== undefined) this.dt = dt;
    addon.setParams(this._h, this.dt, this.G);   // nbody3d.js:470: uniforms cross every frame
    addon.step(this._h, 1);
    return this;
  }

  simulate(nSteps, dt) {
    this._need();
    if (dt !== undefined) this.dt = dt;
    addon.setParams(this._h, this.dt, this.G);
    addon.step(this._h, nSteps >>> 0);
    return this;
  }

  /** util.js:56-64 toggleSim: pause saves dt and zeroes it; a second call restores it. */
  togglePause() {
    if (this._oldDt) { this.dt = this._oldDt; this._oldDt = null; }
    else { this._oldDt = this.dt; this.dt = 0; }
    return this.dt;
  }

  /** util.js:37-46: moving the dt slider while paused changes the saved value only. */
  setDt(newDt) {
    if (this._oldDt) this._oldDt = newDt; else this.dt = newDt;
  }

  sync() { this._need(); addon.sync(this._h); return this; }

  /** util.js:163-178: fresh copies {bodies, vel, accel}. */
  read() {
    this._need();
    const T = this.ArrayType, len = 4 * this.nBodies;
    const out = { bodies: new T(len), vel: new T(len), accel: new T(len) };
    addon.download(this._h, out.bodies, out.vel, out.accel);
    return out;
  }

  readBodies() {
    this._need();
    const b = new this.ArrayType(4 * this.nBodies);
    addon.download(this._h, b, null, null);
    return b;
  }

  /** util.js:230-244: write the three arrays into the EXISTING buffers (N must match). */
  restore(state) {
    this._need();
    addon.upload(this._h, this._coerce(state.bodies, 'bodies'), this._coerce(state.vel, 'vel'),
      state.accel ? this._coerce(state.accel, 'accel') : null);
    return this;
  }

  /** util.js:186-201 schema: {bodies, vel, accel, camera, G: log10(G).toFixed(2)}.  dt and N are not saved upstream either.
   *  The engine has no camera (rendering is out of scope): the `camera` object of the checkpoint this state was imported
   *  from -- or one handed in -- is passed through untouched, so a browser -> engine -> browser round trip keeps the view
   *  (util.js:190-199 writes it, :246-256 restores it and tolerates its absence). */
  exportJSON(camera) {
    const s = this.read();
    const out = { bodies: Array.from(s.bodies), vel: Array.from(s.vel), accel: Array.from(s.accel) };
    const cam = camera !== undefined ? camera : this.camera;
    if (cam !== undefined && cam !== null) out.camera = cam;
    out.G = (Math.log(this.G) / Math.LN10).toFixed(2);
    return JSON.stringify(out);
  }

  /** util.js:217-263.  Unlike the reference, G takes effect on the next step (the
   *  reference forgets uni.GValue.set, SURVEY.md §5.4), and a different N re-creates. */
  importJSON(text) {
    const json = typeof text === 'string' ? JSON.parse(text) : text;
    const T = this.ArrayType;
    const state = { bodies: T.from(json.bodies), vel: T.from(json.vel), accel: json.accel ? T.from(json.accel) : null };
    if (!this._h || state.bodies.length !== 4 * this.nBodies) this.init(state); else this.restore(state);
    if (json.G !== undefined && json.G !== null) this.G = Math.pow(10, parseFloat(json.G));
    this.camera = json.camera !== undefined ? json.camera : null;       // kept for exportJSON, never read by the engine
    return this;
  }

  /** Viewer frame feed.  The reference's render pass reads bodyBuffer and velBuffer in place
   *  every frame (nbody3d.js:408-415,482-487; colour from length(vel.xyz), :380).  requestFrame()
   *  enqueues a snapshot behind the steps issued so far and returns at once: the copy to the host
   *  runs on a second stream and does not stall later step() calls. */
  requestFrame() { this._need(); addon.requestFrame(this._h); return this; }

  /** Newest snapshot that has landed: {bodies: Float32Array(4n) x,y,z,m; speed: Float32Array(n)
   *  |v|; step} -- the same two arrays refreshed on every call -- or null (wait === false and no
   *  frame has landed yet). */
  frame(wait) {
    this._need();
    if (!this._frame) this._frame = { bodies: new Float32Array(4 * this.nBodies), speed: new Float32Array(this.nBodies), step: 0 };
    const step = addon.frame(this._h, wait !== false, this._frame.bodies, this._frame.speed);
    if (step < 0) return null;
    this._frame.step = step;
    return this._frame;
  }

  enableTiming(on) { this._need(); addon.enableTiming(this._h, on !== false); return this; }
  stepTimes() { this._need(); return addon.stepTimes(this._h); }
  collectiveInfo() { this._need(); return addon.collectiveInfo(this._h); }
  kernelTimes() { this._need(); return addon.kernelTimes(this._h); }
  variant() { this._need(); return addon.variant(this._h); }
  diagnostics() { this._need(); addon.setParams(this._h, this.dt, this.G); return addon.diagnostics(this._h); }

  destroy() {
    if (this._h) { addon.destroy(this._h); this._h = null; this._frame = null; }
  }
}

/* Module-level instance: the reference keeps its state in module globals
 * (nbody3d.js:2-34), so `init(p); step(dt)` works without constructing anything. */
let current = null;
function init(particles, options) {
  if (current) current.destroy();
  current = new Simulation(options);
  return current.init(particles);
}
function need() { if (!current) throw new Error('call init(particles) first'); return current; }
function step(dt) { return need().step(dt); }
function simulate(nSteps, dt) { return need().simulate(nSteps, dt); }
function read() { return need().read(); }

module.exports = {
  Simulation: Simulation, init: init, step: step, simulate: simulate, read: read,
  load: load, deviceCount: deviceCount, planQuery: planQuery, TILE_SIZE: TILE_SIZE, EPS2: EPS2, DEFAULT_G: DEFAULT_G, DEFAULT_DT: DEFAULT_DT,
  get current() { return current; }, get abiVersion() { return abi; },
};
