'use strict';
/* Writes the initial conditions js/ic.js::galaxies produces for a parameter file of tests/golden
 * (galaxy_ref_params.json, galaxy40002_params.json: seed, galaxy count, body counts, G, sizeFactor) as raw
 * little-endian float32 arrays <out>_bodies0.f32 / <out>_vel0.f32 and prints {n, sha256_bodies0, sha256_vel0}.
 * The random stream is consumed the way the reference's main() does (nbody3d.js:167-177): the galaxy list
 * first, then generateGalaxy.  Used by the tests and by bench.py for the reference's default workload
 * (N = 40,002), whose state is too large to commit.
 *
 *   node nbody3d-webgpu_amd/js/gen_galaxy.js tests/golden/galaxy40002_params.json /tmp/g40002
 */
const fs = require('fs');
const path = require('path');
const crypto = require('crypto');
const ic = require(path.join(__dirname, 'ic.js'));

const gp = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
const out = process.argv[3];
const stream = ic.mulberry32(gp.seed);
const list = ic.galaxySettings(gp.numGalaxies, { random: stream, minBodies: gp.minBodies, maxBodies: gp.maxBodies });
const gal = ic.galaxies(list, { random: stream, G: gp.G, sizeFactor: gp.outerHeight });
const buf = function (a) { return Buffer.from(a.buffer, a.byteOffset, a.byteLength); };
const sha = function (a) { return crypto.createHash('sha256').update(buf(a)).digest('hex'); };
if (out) {
  fs.writeFileSync(out + '_bodies0.f32', buf(gal[0]));
  fs.writeFileSync(out + '_vel0.f32', buf(gal[1]));
}
console.log(JSON.stringify({ n: gal[0].length / 4, sha256_bodies0: sha(gal[0]), sha256_vel0: sha(gal[1]),
                             settings_match: JSON.stringify(list) === JSON.stringify(gp.galaxySettings) }));
