'use strict';
// fluent-ffmpeg-shaped front end for the scale step, over ttscale.node.
//
// The reference builds its command as
//   ffmpeg(input).outputOptions([...]).size('720x1280').on('end', ...).on('error', ...).save(output)
// (scripts/processVideos.js:71-108).  This module accepts the same chain of calls for RAW yuv420p
// files (decode/encode stay with the host codec): codec options are accepted and ignored, the
// geometry calls (.size, .videoFilters) are turned into the filter string libttscale parses.
const fs = require('fs');
const path = require('path');
const native = require(path.join(__dirname, 'ttscale.node'));

class Command {
  constructor(input) {
    this._input = input;
    this._filters = [];
    this._handlers = {};
    this._inSize = null;
  }
  inputOptions(opts) { return this._options(opts); }
  outputOptions(opts) { return this._options(opts); }
  _options(opts) {
    for (const o of [].concat(opts)) {
      const m = /^-(?:s|video_size)\s+(\d+)x(\d+)$/.exec(String(o).trim());
      if (m) this._inSize = [parseInt(m[1], 10), parseInt(m[2], 10)];
      const f = /^-(?:vf|filter:v)\s+(.+)$/.exec(String(o).trim());
      if (f) this._filters.push(f[1].replace(/^"|"$/g, ''));
    }
    return this;
  }
  inputSize(w, h) { this._inSize = [w, h]; return this; }
  duration() { return this; }
  seekInput() { return this; }
  // fluent-ffmpeg: .size('720x1280') with no aspect/pad option emits scale=w=720:h=1280
  size(s) {
    const m = /^(\d+)x(\d+)$/.exec(s);
    if (!m) throw new Error('INVALID_INPUT: size must be WxH');
    this._filters.push(`scale=w=${m[1]}:h=${m[2]}`);
    return this;
  }
  videoFilters(f) { this._filters.push(...[].concat(f)); return this; }
  on(event, fn) { this._handlers[event] = fn; return this; }
  _emit(event, arg) { if (this._handlers[event]) this._handlers[event](arg); }
  save(output) {
    setImmediate(() => {
      try {
        if (!this._inSize) throw new Error('INVALID_INPUT: raw input needs inputSize(w, h) or "-s WxH"');
        const [iw, ih] = this._inSize;
        const vf = this._filters.length ? this._filters.join(',') : 'format=yuv420p';
        const src = fs.readFileSync(this._input);
        this._emit('progress', { percent: 0 });
        const dst = native.runHost(vf, iw, ih, src);
        fs.writeFileSync(output, dst);
        this._emit('progress', { percent: 100 });
        this._emit('end');
      } catch (err) {
        if (this._handlers.error) this._emit('error', err); else throw err;
      }
    });
    return this;
  }
}

function ttscale(input) { return new Command(input); }
ttscale.parseVf = native.parseVf;
ttscale.frameBytes = native.frameBytes;
ttscale.runHost = native.runHost;
module.exports = ttscale;
