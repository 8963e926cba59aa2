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
    this._fps = 30;
    this._seek = 0;
    this._duration = -1;
    this._output = null;
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
  // `-ss` before the input and `-t` (scripts/loopVideos.js:45-46, scripts/processVideos.js:84-92): raw
  // clips carry no timestamps, so the rate comes from inputFPS(); frame i sits at i/fps.
  inputFPS(fps) { this._fps = fps; return this; }
  fpsInput(fps) { return this.inputFPS(fps); }
  seekInput(t) { this._seek = Number(t) || 0; return this; }
  setStartTime(t) { return this.seekInput(t); }
  duration(d) { this._duration = Number(d); return this; }
  setDuration(d) { return this.duration(d); }
  output(path_) { this._output = path_; return this; }
  run() { return this.save(this._output); }
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
        if (!output) throw new Error('INVALID_INPUT: no output path');
        let src = fs.readFileSync(this._input);
        if (this._seek > 0 || this._duration >= 0) {
          const frame = native.frameBytes(iw, ih);
          if (src.length % frame) throw new Error('INVALID_INPUT: input is not a whole number of frames');
          const [num, den] = fpsRational(this._fps);
          const r = native.trimRange(src.length / frame, num, den, this._seek, this._duration);
          src = src.slice(r.first * frame, (r.first + r.count) * frame);
        }
        this._emit('progress', { percent: 0 });
        // no filter: the frames pass through as they are (a trim or a segment extraction)
        const dst = this._filters.length ? native.runHost(this._filters.join(','), iw, ih, src) : src;
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

// 29.97 -> 30000/1001 and the like; whole rates stay whole
function fpsRational(fps) {
  if (Array.isArray(fps)) return fps;
  if (Number.isInteger(fps)) return [fps, 1];
  const ntsc = Math.round(fps * 1001 / 1000);
  if (Math.abs(ntsc * 1000 / 1001 - fps) < 5e-3) return [ntsc * 1000, 1001];
  return [Math.round(fps * 1000), 1000];
}

function ttscale(input) { return new Command(input); }
ttscale.fpsRational = fpsRational;
ttscale.trimRange = native.trimRange;
ttscale.loopOrder = native.loopOrder;
ttscale.parseVf = native.parseVf;
ttscale.frameBytes = native.frameBytes;
ttscale.runHost = native.runHost;
ttscale.terminate = native.terminate;
module.exports = ttscale;
